# A/B of the carried-history layer at small messages: the product build against an older library in tools/exp/_build_old (tools/ab_old_build.sh header: how to build it)
cd $GRAFT_REPO_ROOT
S="10:8000:4:0:0:carried:drop 12:8000:4:0:0:carried:drop 13:8000:4:0:0:carried:drop 16:8000:4:0:0:carried:drop 20:8000:4:0:0:carried:drop"
fmt='import sys, json
for l in sys.stdin:
    r = json.loads(l); print("2^%d: bare %.2f us per message, graph %.2f us per message (%.3f GS/s)" % (r["log2_msg"], r["bare_us_per_msg"], r["graph_us_per_msg"], r["graph_gsps"]))'
for i in 1 2; do
  echo "== product build (the tail of a small call stays in place)"; tests/_build/kpn_tests bench_c2_list $S 2>&1 | grep "^{" | python3 -c "$fmt"
  echo "== old build (tail copied to the other staging buffer every call)"; LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/exp/_build_old tests/_build/kpn_tests bench_c2_list $S 2>&1 | grep "^{" | python3 -c "$fmt"
done
