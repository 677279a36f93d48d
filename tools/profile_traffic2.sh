#!/bin/bash
# tools/profile_traffic2.sh <rNN> — FETCH_SIZE / WRITE_SIZE (separate --pmc passes) for the wbfm and spectrum benches; the
# spectrum kernel reads every input byte exactly once with the same 2-byte typed loads the wbfm kernel uses, so its known byte
# count calibrates FETCH_SIZE for that access width (guide: "calibrate on a known byte count in your own access pattern").
set -u
TAG=${1:-r01}; OUT=$PWD/gpurun_out/traffic2_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for wl in wbfm spectrum; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --output-format csv --pmc $c -d "$OUT/${wl}_$c" -o pmc -- python3 bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/${wl}_$c.log" 2>&1
  done
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
def mean(wl, c, key):
    v = []
    for fn in glob.glob(os.path.join(out, "%s_%s" % (wl, c), "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            if key in row["Kernel_Name"] and row["Counter_Name"] == c: v.append(float(row["Counter_Value"]))
    return sum(v) / max(len(v), 1)
sp_f, sp_w = mean("spectrum", "FETCH_SIZE", "k_spectrum"), mean("spectrum", "WRITE_SIZE", "k_spectrum")
wb_f, wb_w = mean("wbfm", "FETCH_SIZE", "k_wbfm"), mean("wbfm", "WRITE_SIZE", "k_wbfm")
sp_known = 256 * 234 * 1024 * 2.0                      # bytes the spectrum kernel reads, each exactly once
factor = sp_known / (sp_f * 1024.0)
res = {"calibration": {"kernel": "k_spectrum<10>", "known_read_bytes": sp_known, "FETCH_SIZE_KiB_raw": sp_f, "bytes_per_FETCH_SIZE_byte": factor,
                       "note": "2-byte typed buffer loads (buffer_load_format_xy, 8_8 USCALED), coalesced across the wave"},
       "spectrum": {"kernel_name": "k_spectrum<10>", "FETCH_SIZE_KiB_raw": sp_f, "WRITE_SIZE_KiB_raw": sp_w,
                    "hbm_bytes_per_launch": factor * sp_f * 1024.0 + sp_w * 1024.0},
       "wbfm": {"kernel_name": "wbfm-fused (k_wbfm_fused<8,10>)", "FETCH_SIZE_KiB_raw": wb_f, "WRITE_SIZE_KiB_raw": wb_w,
                "hbm_bytes_per_launch": factor * wb_f * 1024.0 + wb_w * 1024.0}}
json.dump(res, open(os.path.join(out, "traffic2_%s.json" % tag), "w"), indent=1); print(json.dumps(res))
PY
