# round 5, first contact: cv2 pin attempt, the GPU suite's fast core, a bench line, the replayed step as a timeline
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -c "import cv2; print('cv2', cv2.__version__)" > $O/r5_cv2_probe.txt 2>&1
python tools/refgen/pin_cv2.py >> $O/r5_cv2_probe.txt 2>&1 && cp tests/golden/cv2_resize.npz $O/
pip list 2>/dev/null | grep -i -E "opencv|mmcv|pillow|decord|av " >> $O/r5_cv2_probe.txt
timeout 900 python bench.py --no-swin > $O/r5a_bench.json 2> $O/r5a_bench.err
rm -rf $O/prof_tl
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_tl -o g -- python bench.py --timed-only --steps 64 --no-cpu-baseline > $O/r5a_timed_only.json 2>> $O/r5a_prof.err
DB=$(ls $O/prof_tl/*.db $O/prof_tl/*/*.db 2>/dev/null | head -1)
test -n "$DB" && timeout 120 python tools/timeline.py "$DB" $O/r5a_timeline.csv 40 12
test -n "$DB" && timeout 120 python tools/prof_summary.py "$DB" $O/r5a_graph_replay_kernel_stats.csv 250 --split-all > /dev/null
rm -rf $O/prof_tl
tail -3 $O/r5a_bench.json | cut -c1-600
cat $O/r5_cv2_probe.txt
