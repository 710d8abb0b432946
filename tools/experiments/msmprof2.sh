# usage: msmprof2.sh LOGN "ENV=.. ENV=.." ...   one profile per env set
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/msmprof; mkdir -p $out
L=$1; shift
i=0
for E in "$@"; do
  i=$((i+1))
  ( export $E; rocprofv3 --kernel-trace -d $out/q_$i -o m -- ./tools/h2bench msmt $L 254 2 > $out/h2bench_q$i.txt 2>/dev/null )
  echo "#### $E" > $out/q${i}_split.txt
  grep msmt $out/h2bench_q$i.txt >> $out/q${i}_split.txt
  python3 tools/experiments/split_summary.py "$(find $out/q_$i -name '*results.db' | head -1)" | sed -n '/== after/,$p' | grep -v " 0\.[01]%" >> $out/q${i}_split.txt
done
rm -rf $out/q_*
