for cfg in "32768 2" "32768 1" "32768 0" "49152 1" "24576 1" "32768 2"; do set -- $cfg
LPX_KD_HAND=$1 LPX_KD_EXTRA=$2 python3 bench.py --workload ${W:-synth5m} --no-cpu-baseline --no-latency --no-inflight --no-sub --steps 3 --warmup 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('${W:-synth5m} hand $1 extra $2', d['value'], d['ms_per_step'], 'kd alone', d['roofline']['stage_ms_per_launch_alone']['kd_build'])"
done
