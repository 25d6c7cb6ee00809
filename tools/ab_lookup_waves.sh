for w in 6 7 8; do
  PEMAP_LOOKUP_WAVES=$w timeout -k 10 200 python bench.py --steps 12 --warmup 2 --no-cpu --no-secondary --no-pecaller --realistic-steps 0 > gpurun_out/r3_abw_$w.log 2> gpurun_out/r3_abw_$w.err
  python3 -c "
import json;d=json.loads(open('gpurun_out/r3_abw_$w.log').read().strip().splitlines()[-1]);print('waves $w', {k:d[k] for k in ['value','ms_per_step','resident_value','resident_ms_per_step']}, d['roofline']['avg_launch_ms'])"
done
