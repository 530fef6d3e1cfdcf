for ch in 390 521 625 781 1042 1563; do
  python bench.py --steps 20 --warmup 3 --no-cpu --chunk $ch 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chunk', $ch, d['ms_per_step'], d['kernel_ms']['fwdbwd'], d['config'].get('speculative_boundaries'))"
done
for w in 160 200 240 288; do
  BHMM_AMD_SPEC_W=$w python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('W', $w, d['ms_per_step'], d['kernel_ms']['fwdbwd'], d['config'].get('speculative_boundaries'))"
done
