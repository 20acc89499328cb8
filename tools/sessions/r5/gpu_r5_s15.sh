#!/bin/bash
ulimit -c 0
# Round 5, session 15: the launch-mode table of tools/gpu_final.sh again on one box (the round-end session lost its whole-step-capture row: that run
# died without a JSON line and its stderr was discarded; it did not reproduce in r5s14) - stderr kept this time
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
B2="python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3"
i=0
( for g in on piecewise "piecewise --pace stream --overlap-optimizer off" off; do i=$((i+1)); timeout 300 $B2 --graph $g --force-collectives --bucket-timeline 2> $out/mode_$i.err | python -c "
import sys, json
try:
    d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
except StopIteration:
    print('graph=$g: NO JSON LINE (see mode_$i.err)'); sys.exit(0)
bt = d.get('bucket_timeline') or {}
print('graph=$g, gradient exchange through a 1-rank RCCL group:', d['value'], 'images/s', d['ms_per_step'], 'ms/step, mode', d['config']['graph_mode'], 'pace', d['config'].get('collective_pace'), 'overlapped optimizer', d['config'].get('overlapped_optimizer'), '| eager timeline step: backward', bt.get('backward_ms'), 'ms, exchange finished', bt.get('exchange_finished_ms'), 'ms, exposed', bt.get('exposed_communication_ms'), 'ms,', len(bt.get('buckets', [])), 'buckets')
if '$g' == 'off':
    for r in bt.get('buckets', []): print('   ', r)
"; done
  timeout 300 $B2 --graph on 2> $out/mode_nocoll.err | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
print('graph=on, no collectives (the single-GPU default):', d['value'], 'images/s', d['ms_per_step'], 'ms/step')" ) > $out/launch_modes_one_rank_rccl.txt
grep "^graph=" $out/launch_modes_one_rank_rccl.txt | cut -c1-230
for f in $out/mode_*.err; do if grep -q "Traceback\|Error\|fault" $f; then echo "== $f"; grep -v "transformers\|GenerationMixin\|trust_remote\|owner of the model\|embeddings will be" $f | tail -n 15 | cut -c1-300; fi; done
