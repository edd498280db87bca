# A/B of two builds of the tuning flavour on one box: libtlxmi_tune_old.so (copied aside before a change) vs the current one;
# alternating runs of tools/ab_graph.py (hipGraph replay of a whole forward).  usage: ab_oldnew.sh <workload> <batch> [VAR vals]
wl=${1:-vit_b16}; bs=${2:-256}; var=${3:-TLXMI_NOP}; vals=${4:-0}
for rep in 1 2; do
  TLXMI_TUNE_LIB=$PWD/tlxcv_amd/libtlxmi_tune_old.so python tools/ab_graph.py $var $vals $wl $bs 2>&1 | grep batch | sed 's/^/old: /'
  python tools/ab_graph.py $var $vals $wl $bs 2>&1 | grep batch | sed 's/^/new: /'
done
