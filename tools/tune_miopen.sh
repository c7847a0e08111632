#!/bin/bash
# One-time tuning of the vendor convolutions for the bench shapes (run on an MI355X through gpurun from the repo root):
# bench.py --miopen-search makes PyTorch call MIOpen's Find for every convolution shape of a TTA step and of the Dice pass
# (4 x 3 x 800 x 800 fp32; ~100 problem configurations, forward / backward-data / backward-weights), MIOpen times every applicable
# solver and records the results in its user find-db / perf-db: two text files under MIOPEN_USER_DB_PATH.  Copy them into
# ttdg-mgm_amd/miopen_db/ (the package points MIOPEN_USER_DB_PATH there at import): from then on MIOpen's immediate mode - what
# PyTorch uses with cudnn.benchmark False - returns the measured-fastest solver for these shapes instead of its heuristic's choice.
# Measured: 400 s of search once; 84.0 -> 85.8 - 86.3 adapted images/s; a run with the db and an EMPTY kernel cache starts as fast
# as without (the chosen solvers' kernels are in MIOpen's system kernel database).
exec < /dev/null
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/miopen_db}
mkdir -p "$OUT"
cp -n ttdg-mgm_amd/miopen_db/*.txt "$OUT"/ 2>/dev/null      # keep the records already shipped (MIOpen appends: both memory layouts stay covered)
python bench.py --steps 2 --warmup 1 --no-ab --no-cpu-baseline > /dev/null 2>&1      # fits / caches the checkpoint outside the search
S=$(date +%s)
MIOPEN_USER_DB_PATH=$PWD/$OUT timeout 1800 python bench.py --miopen-search --steps 4 --warmup 2 --no-ab --no-cpu-baseline > "$OUT/search_bench.json" 2> "$OUT/search.err"
echo "search rc=$? took $(( $(date +%s) - S )) s"
ls -la "$OUT"
