cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in "" "--streams 1"; do
  tag=$( [ -z "$mode" ] && echo main || echo streams1 )
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --no-cpu-baseline --no-grouped --steps 4 --warmup 1 $mode > $R/gpurun_out/prof_$tag.json 2> $R/gpurun_out/prof_$tag.err
  find $R/gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete      # tens of MB; the stats summary is what is kept
  tail -c 600 $R/gpurun_out/prof_$tag.json | head -c 200; echo
done
