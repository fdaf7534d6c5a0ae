# L2 counters of every launch of the operand-feed probe (tools/micro/feed_probe.hip), one line per launch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mp_feed
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $O -- $R/tools/micro/bin/feed_probe > $O.log 2>&1 || exit 1
python3 - $O <<'PY' > $R/gpurun_out/mp_feed.txt
import csv, glob, sys, collections
rows = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = rows.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"][:12]})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
for i, d in sorted(rows.items()):
    if "TCC_READ_sum" not in d: continue
    rd, ea = d["TCC_READ_sum"], d["TCC_EA0_RDREQ_sum"]
    lat = d["TCP_TCC_READ_REQ_LATENCY_sum"] / max(d["TCP_TCC_READ_REQ_sum"], 1)
    print(f"{i:3d} {d['k']:12s} L2 reads {rd:10.3e}  fabric reads {ea:10.3e} ({100*ea/max(rd,1):5.1f} %)  mean L1->L2 latency {lat:6.0f}  "
          f"in flight per CU {d['TCP_TCC_READ_REQ_LATENCY_sum'] / (d['GRBM_GUI_ACTIVE'] / 8 * 256):6.1f}")
PY
