#!/bin/bash
# Build the C-ABI library of another git revision next to the in-tree one (same-box kernel A/B through FALNET_LIB):
#   tools/build_alt.sh <rev> <tag>  ->  fal_net_amd/libfalnet_hip_<tag>.so   (the revision must have the same C-ABI structs)
rev=$1; tag=$2; tmp=$(mktemp -d); root="$(cd "$(dirname "$0")/.." && pwd)"
git -C "$root" archive "$rev" fal_net_amd/csrc fal_net_amd/_build.py include | tar -x -C "$tmp"
srcs=$(python - "$tmp/fal_net_amd/_build.py" <<'PY'
import re, sys
print(" ".join(re.findall(r'"([^"]+\.(?:hip|cpp))"', re.search(r"SOURCES = \[(.*?)\]", open(sys.argv[1]).read(), re.S).group(1))))
PY
)
objs=""
for f in $srcs; do
  x=""; [[ $f == *.hip ]] && x="-x hip"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $x -c "$tmp/fal_net_amd/csrc/$f" -o "$tmp/$f.o" &
  objs="$objs $tmp/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/fal_net_amd/libfalnet_hip_$tag.so" $objs && echo "built libfalnet_hip_$tag.so"
rm -rf "$tmp"
