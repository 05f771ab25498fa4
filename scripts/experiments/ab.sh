# A/B of library variants on one box, round robin: ab.sh <out> <rounds> <shape> <configs> <variant>...   ("main" = the product library)
out=$1; rounds=$2; shape=$3; cfgs=$4; shift 4
mkdir -p $(dirname $out); : > $out
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    lib=$PWD/surge_amd/libsurge_replay_$v.so; [ "$v" = main ] && lib=$PWD/surge_amd/libsurge_replay.so
    echo "== round $r variant $v" >> $out
    SURGE_REPLAY_LIB=$lib NOCHECK=1 SHAPES=$shape CONFIGS=$cfgs FOLDS=${FOLDS:-20} timeout 600 python scripts/experiments/tiled_probe.py 2>&1 | grep -v amdgpu.ids >> $out
  done
done
python3 - $out <<'PY'
import re, sys, collections
cur=None; d=collections.defaultdict(list)
for line in open(sys.argv[1]):
    m=re.match(r"== round \d+ variant (\S+)", line)
    if m: cur=m.group(1); continue
    m=re.match(r"\s+(auto \(CSR kernel\)|tiled subs=\d waves/CU=\d+).*med ([\d.]+)", line)
    if m: d[(cur,m.group(1))].append(float(m.group(2)))
for k,v in sorted(d.items()):
    v2=sorted(v); print(f"{k[0]:10s} {k[1]:28s} n={len(v)} median-of-medians {v2[len(v2)//2]:.4f}  all {' '.join(f'{x:.3f}' for x in v)}")
PY
