export VPX_LIB=build/libvpx_ablate.so
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for g in 512 384 256 1024; do
export VPX_C1_GRID=$g
rm -rf gpurun_out/kq; mkdir -p gpurun_out/kq
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kq -- python3 bench.py --model predrnn-pp --mode infer --steps 3 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
echo "== grid $g"; python3 tools/kstats.py gpurun_out/kq 3 12 | grep c1_kernel
done
rm -rf gpurun_out/kq
