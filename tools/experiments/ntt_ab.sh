# same-box A/B of two builds of the library (LD_LIBRARY_PATH wins over the binary's RUNPATH)
# usage: $0 <directory holding the OLD libhalo2_hip.so>
OLD=${1:?directory of the old libhalo2_hip.so}
for round in 1 2 3; do
  echo "== new"; ./tools/h2bench ntt 24 5; ./tools/h2bench ntt 24 20; ./tools/h2bench ntt 25 10; ./tools/h2bench ntt 22 20
  echo "== old"; LD_LIBRARY_PATH=$OLD ./tools/h2bench ntt 24 5; LD_LIBRARY_PATH=$OLD ./tools/h2bench ntt 24 20;  LD_LIBRARY_PATH=$OLD ./tools/h2bench ntt 25 10; LD_LIBRARY_PATH=$OLD ./tools/h2bench ntt 22 20
done
