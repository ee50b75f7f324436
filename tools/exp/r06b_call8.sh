O=gpurun_out/r06b; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/gputests2.txt 2>&1; tail -6 $O/gputests2.txt | head -3
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/scene_prof -o scene -- python $GRAFT_REPO_ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 50 --batched-only > $GRAFT_REPO_ROOT/$O/scene_under_trace.json 2> $GRAFT_REPO_ROOT/$O/scene_prof.err
cd $GRAFT_REPO_ROOT; f=$(find $O/scene_prof -name "*kernel_stats.csv" | head -1); cp $f $O/scene_kernel_stats_batch_dyn.csv; rm -rf $O/scene_prof; head -8 $O/scene_kernel_stats_batch_dyn.csv | cut -c1-180
