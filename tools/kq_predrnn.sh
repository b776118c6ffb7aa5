cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for m in infer train; do
rm -rf gpurun_out/kq; mkdir -p gpurun_out/kq
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kq -- python3 bench.py --model predrnn-pp --mode $m --steps 3 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
echo "== predrnn $m"; python3 tools/kstats.py gpurun_out/kq 3 10
done
rm -rf gpurun_out/kq
