# same-box A/B of two builds of the library (LD_LIBRARY_PATH wins over the binary's RUNPATH)
for round in 1 2 3; do
  echo "== new"; ./tools/h2bench ntt 24 5; ./tools/h2bench ntt 24 20; ./tools/h2bench ntt 25 10; ./tools/h2bench ntt 22 20
  echo "== old"; LD_LIBRARY_PATH=tools/experiments/oldlib ./tools/h2bench ntt 24 5; LD_LIBRARY_PATH=tools/experiments/oldlib ./tools/h2bench ntt 24 20;  LD_LIBRARY_PATH=tools/experiments/oldlib ./tools/h2bench ntt 25 10; LD_LIBRARY_PATH=tools/experiments/oldlib ./tools/h2bench ntt 22 20
done
