#!/bin/bash
# ONE command for the first session on an 8-GPU MI355X node (nobody has run this path on more than one device yet):
#     bash tools/sessions/r5/scale8.sh [outdir] [--rehearsal]
# N in {1, 2, 4, 8} x the arms DESIGN.md section 6 lists, same workload as the driver's SCALE run (config B, per-GPU batch 32, weak scaling), one JSON
# line per run under <outdir>/, and at the end one table: images/s, ms/step, weak-scaling efficiency against this script's own N = 1 line, the launch
# mode every rank agreed on, exposed communication of the eager timeline step and the last six buckets' ready / done times.
#   arms:   default            piecewise replay, host-paced collectives, AdamW beside the backward, resampler layer by layer, 4 layers per segment
#           stream             --pace stream --overlap-optimizer off       (collectives ordered by cross-stream waits)
#           seg12              --segment-layers 12 --wgrad-group 12 --kv-group 12   (fewer, larger exchanges)
#           stack-resampler    --resampler-layerwise off                  (one 126 MB resampler bucket at the end of backward)
#           ch8                --rccl-channels 8                          (RCCL on fewer CUs beside the one-tile-per-CU GEMM launches)
#           full-graph         --graph on                                 (collectives captured in ONE graph)
#           sharded            --optimizer sharded                        (reduce-scatter -> sharded AdamW -> all-gather per bucket)
#           eager              --graph off
# --rehearsal: every rank on device 0 with a gloo exchange (bench.py --shared-gpu-rehearsal) - checks the plumbing of THIS script on a one-GPU box
# (N in {1, 2} and three arms only; its throughput is not a measurement).
ulimit -c 0
R=$(cd "$(dirname "$0")/../../.." && pwd)
out=${1:-$R/gpurun_out/scale8}; mkdir -p $out
reh=""; NS="1 2 4 8"; ARMS="default stream seg12 stack-resampler ch8 full-graph sharded eager"
if [ "$2" == "--rehearsal" ]; then reh="--shared-gpu-rehearsal"; NS="1 2"; ARMS="default stack-resampler eager"; fi
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
arm_flags() {
  case $1 in
    default) echo "";;
    stream) echo "--pace stream --overlap-optimizer off";;
    seg12) echo "--segment-layers 12 --wgrad-group 12 --kv-group 12";;
    stack-resampler) echo "--resampler-layerwise off";;
    ch8) echo "--rccl-channels 8";;
    full-graph) echo "--graph on";;
    sharded) echo "--optimizer sharded";;
    eager) echo "--graph off";;
  esac
}
port=29500
for n in $NS; do
  for arm in $ARMS; do
    if [ $n == 1 ] && [ $arm != default ]; then continue; fi          # N = 1: the driver's own line (one graph, no collectives)
    if [ -n "$reh" ] && [ $arm == full-graph ]; then continue; fi
    f=$out/n${n}_$arm.json
    flags="$(arm_flags $arm) --no-cpu-baseline --caption-tokens 0 --companions off --profile-steps 0 --steps 20 --warmup 5 --bucket-timeline $reh"
    if [ $n == 1 ]; then
      timeout 600 python bench.py --gpus 1 $flags > $f 2> ${f%.json}.err
    else
      port=$((port + 1))
      timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n $flags > $f 2> ${f%.json}.err
    fi
    echo "n=$n arm=$arm rc=$?"
  done
done
python - $out <<'P'
import glob, json, os, sys
out = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(out, "n*_*.json"))):
    n, arm = os.path.basename(f)[1:-5].split("_", 1)
    try:
        rows[(int(n), arm)] = next(json.loads(l) for l in reversed(open(f).read().strip().splitlines()) if l.startswith("{"))
    except StopIteration:
        rows[(int(n), arm)] = None
base = rows.get((1, "default"))
print(f"{'N':>2} {'arm':16} {'images/s':>10} {'ms/step':>8} {'eff':>6}  mode / pace / overlapped AdamW / layerwise | backward, exchange finished, exposed (ms, eager timeline step) | last buckets")
for (n, arm), d in sorted(rows.items()):
    if d is None:
        print(f"{n:>2} {arm:16} no JSON line (see the .err file)")
        continue
    c, bt = d["config"], d.get("bucket_timeline") or {}
    eff = d["value"] / (n * base["value"]) if base else float("nan")
    last = "; ".join(f"{b['bucket'][-36:]} {b['mb']:.0f}MB {b['ready_ms']:.1f}->{b['done_ms']:.1f}" for b in bt.get("buckets", [])[-6:])
    print(f"{n:>2} {arm:16} {d['value']:>10.1f} {d['ms_per_step']:>8.2f} {eff:>6.3f}  {c['graph_mode']} / {c.get('collective_pace')} / {c.get('overlapped_optimizer')} / {c.get('resampler_layerwise')} | "
          f"{bt.get('backward_ms')}, {bt.get('exchange_finished_ms')}, {bt.get('exposed_communication_ms')} | {last}")
P
