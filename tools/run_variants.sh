# development: the 40-step bench line under library switches (bench.py --debug "k=v,..."), one box:
#   bash tools/run_variants.sh "" "4=2" "6=2,4=1"
for dbg in "${@:-}"; do
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --debug "$dbg" > gpurun_out/var.json 2> gpurun_out/var.err
  python - "$dbg" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/var.json').read().strip().splitlines()[-1]); r=d['roofline']
print("debug %-10s ms/step %.4f  fused kernel %.2f us p50 %.2f frac %.3f | gather operator %.2f us frac %.3f" % (sys.argv[1], d['ms_per_step'], r['avg_launch_us'], r['launch_us']['p50'], r['frac'], r['gather_operator']['avg_launch_us'], r['gather_operator']['frac']))
PY
done
