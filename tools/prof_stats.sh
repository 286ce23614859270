#!/bin/bash
# dev helper: rocprofv3 kernel stats of one command.  usage: tools/prof_stats.sh <tag> <python args...>
# writes gpurun_out/prof_<tag>/ and prints the condensed table
tag=$1; shift
mkdir -p gpurun_out
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 "$@" > gpurun_out/prof_$tag.log 2>&1
echo "rocprof rc=$?"
python3 tools/summarize_prof.py gpurun_out/prof_$tag | grep -v -e rocprim -e "k_mark\|k_compact\|k_rowptr\|k_degree\|k_fill\|k_gcn_norm\|k_bip\|k_rel_keys\|k_seg_keys\|k_lower\|k_item\|k_rank\|k_deal\|k_slot\|k_indegree\|k_i32" | head -${TOPN:-14}
