cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for f in 0 1; do
echo "fuse_first=$f: $(CDN_HEADS_FUSE_FIRST=$f python tools/heads_bench.py 2>/dev/null | cut -c1-60) $(CDN_HEADS_FUSE_FIRST=$f python tools/e2e_native_bench.py --graph 2>/dev/null | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_batch"])')"
done
done
