cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for side in trivial backbone forward; do
  echo "== side: $side"
  SPLIT_SIDE=$side python -X faulthandler tools/experiments/split_batch_probe.py 2>&1 | grep -a "SPLIT\|Fatal\|File \"/root/repo" | head -5
done
