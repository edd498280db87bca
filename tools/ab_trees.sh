#!/bin/bash
# Same-box A/B of two source trees (the working tree against a `git worktree` copy of an earlier commit built beside it, e.g. ab_old/):
# alternating runs of tools/ab_graph.py (hipGraph replay of a whole forward) in each.  usage: ab_trees.sh <old tree> <workload> <batch> [reps]
old=${1:-ab_old}; wl=${2:-vit_b16}; bs=${3:-256}; reps=${4:-2}
for rep in $(seq $reps); do
  (cd $old && python3 tools/ab_graph.py TLXMI_NOP 0 $wl $bs 2>&1 | grep batch | sed 's/^/old: /')
  python3 tools/ab_graph.py TLXMI_NOP 0 $wl $bs 2>&1 | grep batch | sed 's/^/new: /'
done
