#!/bin/bash
# A/B on ONE box (devices differ by several %): bench.py forward time under environment toggles, interleaved rounds.
#   bash tools/ab_bench.sh "VTI_NO_PK2=1" "VTI_NO_BNECK=1" ...   (first arm is always the default plan)
arms=("A=default" "$@")
for round in 1 2; do
  for arm in "${arms[@]}"; do
    env $arm python bench.py --dtype ${DT:-h2} --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-line --no-fp16-line --no-host-fed --parity-frames 0 --preheat 0.5 > /tmp/ab.json 2>/dev/null
    python - "$arm" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(f"{sys.argv[1]:28s} fwd {d['roofline']['avg_ms']:.4f} ms  iso {d['roofline']['isolated']['avg_ms']:.4f}  step {d['ms_per_step']:.4f}  post {d['stage_ms']['nms+masks+scale_boxes']:.4f}  fps {d['value']:.0f}")
PY
  done
done
