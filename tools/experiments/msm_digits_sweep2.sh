cd "$GRAFT_REPO_ROOT"
for spec in "19 14 15 16" "21 13 14" "23 12 13" "20 14 15" "24 12 13"; do
  set -- $spec; L=$1; shift
  echo "== log_n $L default"; ./tools/h2bench msmt $L 254 5 2>/dev/null | grep msmt
  for D in "$@"; do echo "-- digits $D"; H2_MSM_TABLE_FORCE=1 H2_MSM_TABLE_DIGITS=$D ./tools/h2bench msmt $L 254 5 2>/dev/null | grep msmt; done
done
