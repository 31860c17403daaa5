#!/bin/bash
# host-side profile of the unmodified reference loop (cfg2, autograph): where do the ~1.9 ms per step go?
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
timeout 300 python -m cProfile -o /tmp/cfg2.prof bench.py --config cfg2 --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs > gpurun_out/v_cfg2.json 2> gpurun_out/v_cfg2.err < /dev/null
python - <<'PY' > gpurun_out/v_cfg2_profile.txt 2>&1
import pstats
p = pstats.Stats('/tmp/cfg2.prof')
p.sort_stats('cumulative').print_stats(70)
p.sort_stats('tottime').print_stats(40)
PY
python tools/bench_brief.py < gpurun_out/v_cfg2.json
