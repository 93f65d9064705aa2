echo "## --cases 1200 --seed 51 (both optimizers, 4..40 tensors, every option)"
timeout 1500 python tools/fuzz_gpu.py --cases 1200 --seed 51 2>&1 | tail -1
echo "## --cases 500 --seed 52 --which fw --nmin 40 --nmax 130 --dims two"
timeout 1500 python tools/fuzz_gpu.py --cases 500 --seed 52 --which fw --nmin 40 --nmax 130 --dims two 2>&1 | tail -1
echo "## TNCO_HIP_FW_BIG=1 --cases 300 --seed 53 --which fw --nmin 20 --nmax 90 --dims two"
TNCO_HIP_FW_BIG=1 timeout 1500 python tools/fuzz_gpu.py --cases 300 --seed 53 --which fw --nmin 20 --nmax 90 --dims two 2>&1 | tail -1
echo "## --cases 300 --seed 54 --which im --nmin 150 --nmax 400 (deep / large trees through build_kernel)"
timeout 1500 python tools/fuzz_gpu.py --cases 300 --seed 54 --which im --nmin 150 --nmax 400 2>&1 | tail -1
echo "## python tools/fuzz_forms.py --cases 40 --seed 9"
timeout 1500 python tools/fuzz_forms.py --cases 40 --seed 9 2>&1 | tail -3
