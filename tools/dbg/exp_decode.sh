#!/bin/bash
# usage: exp_decode.sh "<bench.py flags>" ...   -> decode_ms / us per token per set of flags ("" = the product configuration;
# A/B switches: --no-graphs, --no-ln-fusion)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for cfg in "$@"; do
  out=$(python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --pipelines 1 --decode-groups 1 $cfg 2>/dev/null | tail -1)
  echo "[$cfg] => $(echo "$out" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); p=j["phases_ms"]; print("value %.0f  ms/step %.1f  enc %.1f  dec %.2f  us/tok %.1f  gemm %.0f TF" % (j["value"], j["ms_per_step"], p["encoder_ms"], p["decode_ms"], p["decode_ms"]*1e3/j["decode_steps"], j["roofline"]["achieved"]))')"
done
