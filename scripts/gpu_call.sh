set -u
OUT=gpurun_out/r4c49; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
echo "# tests/probes/bwd_kernel_ab.py and fwd_kernel_ab.py on one MI355X: the compiler-allocated GEMM kernels (PLM_*_KERNEL=0)"
echo "# against the assembly-loop kernels (=1) in the same process; HIP-event times, gradients compared bit for bit"
for shape in "50000 300" "100000 300" "50000 500" "20000 200"; do
  echo "## N L = $shape, three digit planes"
  timeout 200 python tests/probes/bwd_kernel_ab.py $shape 2>&1 | tail -3
  timeout 200 python tests/probes/fwd_kernel_ab.py $shape 2>&1 | tail -3
done
echo "## N L = 50000 300, four digit planes (accurate evaluations)"
timeout 200 python tests/probes/bwd_kernel_ab.py 50000 300 4 2>&1 | tail -3
} > $OUT/kernel_ab.txt 2>&1
cat $OUT/kernel_ab.txt
