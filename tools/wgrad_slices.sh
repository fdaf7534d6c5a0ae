# Times the weight-gradient kernel with the split over M forced to 1..8 slices (PDA_WGRAD_SLICES).
R=$GRAFT_REPO_ROOT
for s in 2 3 4 5 6 8; do
  echo "== slices $s"
  PDA_WGRAD_SLICES=$s timeout -k 10 100 python3 $R/tools/wgrad_probe.py || exit 1
done
