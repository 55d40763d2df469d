#!/bin/bash
# A/B of frame time between two environments, alternating runs on the same box.
# usage: ab_frames.sh "ENV_A=.." "ENV_B=.." [reps] [frame_prof args]
A="$1"; B="$2"; reps=${3:-3}; shift 3
for i in $(seq $reps); do
  for e in "$A" "$B"; do
    r=$(env $e python scripts/frame_prof.py --frames 100 "$@" 2>/dev/null | grep FRAMES | awk '{print $3}')
    echo "$e $r"
  done
done | sort | awk '{k=$1; s[k]+=$2; n[k]++; v[k]=v[k]" "$2} END {for (k in s) printf "%-28s mean %.3f ms/frame  (%s )\n", k, s[k]/n[k], v[k]}'
