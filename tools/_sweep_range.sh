for lp in 24 22 20; do for lr in 9 10 11 12 13 14; do
echo "log_points=$lp log_range=$lr"; ZKGPU_EXP_LOG_RANGE=$lr timeout -k 10 120 python bench.py --workload msm --log-points $lp --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['msm_phase_ms'])" || exit 1
done; done
