cd $GRAFT_REPO_ROOT
python -X faulthandler tools/experiments/split_batch_probe2.py 2>&1 | grep -a "SPLIT\|Fatal\|Error\|File \"/root/repo" | head -8
