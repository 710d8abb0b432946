# same-box A/B of two builds of the library (LD_LIBRARY_PATH wins over the binary's RUNPATH)
for round in 1 2; do
  for lib in new old; do
    echo "== $lib"
    if [ $lib = old ]; then export LD_LIBRARY_PATH=tools/experiments/oldlib; else unset LD_LIBRARY_PATH; fi
    ./tools/h2bench msmt 20 254 10 | cut -c1-200; ./tools/h2bench msmt 22 254 5 | cut -c1-200; ./tools/h2bench msmt 24 254 3 | cut -c1-200; ./tools/h2bench msm 18 254 10 | cut -c1-200; ./tools/h2bench msm 20 16 10 | cut -c1-200
  done
done
unset LD_LIBRARY_PATH
python -m pytest tests/test_gpu_msm_table.py tests/test_gpu_parity.py -x -q -m gpu -k "msm" 2>&1 | tail -2
python tools/msm_fuzz.py 60 1 tables 2>&1 | tail -2
