mkdir -p gpurun_out/r04k
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/r04k/prof --output-format csv -- python3 bench.py --integrator directlighting --spp 64 --steps 1 --warmup 1 --no-cpu-baseline --no-spp1024 > gpurun_out/r04k/bench.json 2> gpurun_out/r04k/bench.err
find gpurun_out/r04k/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04k/kernel_trace.csv; rm -rf gpurun_out/r04k/prof
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r04k/kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if r["Kernel_Name"].startswith(("k_trace", "k_rec", "k_gen", "k_film", "void rocprim"))]
for r in sel[-40:]:
    print("%-40s %9.3f ms" % (r["Kernel_Name"][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
