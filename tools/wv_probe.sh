for w in 4 6 7 8; do
PEMAP_LOOKUP_WAVES=$w PEMAP_PIPELINE=2 PEMAP_LIB=/root/repo/pecaller_amd/libpemap_hip.probes.so timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-secondary --no-pecaller > gpurun_out/probe_w$w.log 2>&1 || exit 1
echo "waves $w: $(grep pm_s4_probe gpurun_out/probe_w$w.log | tail -1 | cut -c1-420)"
grep '^{' gpurun_out/probe_w$w.log | tail -1 | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('   lookup ms/step', d['roofline']['kernel_ms']['lookup'], 'step', d['resident_ms_per_step'])"
done
