export H2_MSM_TWO_LEVEL_MIN_HI=2
for L in 16 17 18 19 20 21; do ./tools/h2bench msmt $L 254 2 | grep msmt | cut -c1-150; done
python tools/msm_fuzz.py 60 11 tables 2>&1 | tail -1
python -m pytest tests/test_gpu_msm_table.py -x -q -m gpu 2>&1 | tail -1
unset H2_MSM_TWO_LEVEL_MIN_HI
python -m pytest tests/test_gpu_msm_table.py tests/test_gpu_parity.py -x -q -m gpu -k "msm" 2>&1 | tail -1
