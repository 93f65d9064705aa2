mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( echo "# tools/fuzz_gpu.py on the round-4 library (tnco_hip 0.5), fresh random cases against the oracle"
  echo "## --cases 1200 --seed 41 (both optimizers, 4..40 tensors, every option)"; timeout 1500 python tools/fuzz_gpu.py --cases 1200 --seed 41 2>&1 | tail -4
  echo "## --cases 500 --seed 42 --which fw --nmin 40 --nmax 130 --dims two (hyper-index networks on the one-wavefront re-slice)"; timeout 1500 python tools/fuzz_gpu.py --cases 500 --seed 42 --which fw --nmin 40 --nmax 130 --dims two 2>&1 | tail -4
  echo "## TNCO_HIP_FW_BIG=1 --cases 300 --seed 43 --which fw --nmin 20 --nmax 90 --dims two (its roomier configuration)"; TNCO_HIP_FW_BIG=1 timeout 1500 python tools/fuzz_gpu.py --cases 300 --seed 43 --which fw --nmin 20 --nmax 90 --dims two 2>&1 | tail -4
  echo "## TNCO_HIP_FW_WAVE=0 --cases 300 --seed 44 --which fw --nmin 20 --nmax 90 (walk + full rebuild on the split layout, hyper legs included)"; TNCO_HIP_FW_WAVE=0 timeout 1500 python tools/fuzz_gpu.py --cases 300 --seed 44 --which fw --nmin 20 --nmax 90 2>&1 | tail -4
) > gpurun_out/r04/fuzz.txt 2>&1
cat gpurun_out/r04/fuzz.txt
