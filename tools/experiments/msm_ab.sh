# same-box A/B of two builds of the library (LD_LIBRARY_PATH wins over the binary's RUNPATH)
# usage: $0 <directory holding the OLD libhalo2_hip.so>
OLD=${1:?directory of the old libhalo2_hip.so}
python -m pytest tests/test_gpu_msm_table.py tests/test_gpu_parity.py -x -q -m gpu -k "msm" 2>&1 | tail -1
python tools/msm_fuzz.py 30 17 tables 2>&1 | tail -1
for lib in new old new old; do
    echo "== $lib"
    if [ $lib = old ]; then export LD_LIBRARY_PATH=$OLD; else unset LD_LIBRARY_PATH; fi
    ./tools/h2bench msmt 20 254 5 | grep msmt | cut -c1-160; ./tools/h2bench msmt 24 254 2 | grep msmt | cut -c1-160
done
unset LD_LIBRARY_PATH
