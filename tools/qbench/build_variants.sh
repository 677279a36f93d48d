#!/bin/bash
# tools/qbench/build_variants.sh name=DEFS ... — one qbench binary per variant of the design-Q kernel (csrc/sdrfm_q.hip compiled with the given -D flags), e.g.
#   bash tools/qbench/build_variants.sh prod= base="-DSDRFM_Q_K3=0 -DSDRFM_Q_XORSKIP=0" m2=-DSDRFM_Q_MAGIC=2
# The binaries (qbench_<name>, git-ignored) travel to the GPU box with the snapshot; tools/ab.sh runs them in turn on one box.
cd "$(dirname "$0")" || exit 1
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}
  make -s clean >/dev/null; make -s QDEFS="$defs" qbench 2>&1 | grep -vE 'warning: argument unused|^$' ; mv qbench qbench_$name && echo "built qbench_$name [$defs]"
done
