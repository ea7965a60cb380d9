# usage (through gpurun): bash tools/kernel_times.sh tools/<script>.py [args]  -> per-kernel calls / average ns / %
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
S=$1; shift
cd /tmp && rm -rf /tmp/kt_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_out -o kt -- python $R/$S "$@" > /tmp/kt_run.log 2>&1 || tail -5 /tmp/kt_run.log
F=$(find /tmp/kt_out -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if r and ("k_" in r[0] or r[0] == "Name"):
        print(r[0].replace("void (anonymous namespace)::", "")[:58].ljust(60), r[1].rjust(6), r[3].rjust(12), r[4].rjust(8))
PY
