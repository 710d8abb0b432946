for round in 1 2; do
  for w in 4 5; do
    echo "== H2_MSM_ACC_WAVES=$w"
    H2_MSM_ACC_WAVES=$w ./tools/h2bench msmt 20 254 5 msmt 22 254 3 msmt 24 254 2 | grep msmt | cut -c1-150
  done
done
