#!/bin/bash
ulimit -c 0
# the driver's multi-rank command on a one-GPU box: 2 ranks share device 0, gloo exchange (bench.py --shared-gpu-rehearsal)
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
for extra in "" "--graph on" "--graph off"; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --shared-gpu-rehearsal $extra > $out/two_ranks.out 2> $out/two_ranks.err; echo "rc=$? [$extra]"
grep -c "^{" $out/two_ranks.out
python - <<P
import json
lines=[l for l in open("$out/two_ranks.out").read().strip().splitlines() if l.startswith("{")]
d=json.loads(lines[-1]); c=d["config"]
print(len(lines), "JSON line(s):", d["value"], d["unit"], "n_gpus", d["n_gpus"], d["ms_per_step"], "ms/step, mode", c.get("graph_mode"), "collectives", c.get("collectives"), "pace", c.get("collective_pace"), "overlapped optimizer", c.get("overlapped_optimizer"), "loss", c.get("loss"), c.get("rehearsal"), "| global batch", c.get("global_batch"), "| host", c.get("piecewise_host_ms_per_step"))
print("bucket_timeline" in d, (d.get("bucket_timeline") or {}).get("exposed_communication_ms"))
P
tail -3 $out/two_ranks.err | cut -c1-300
done
