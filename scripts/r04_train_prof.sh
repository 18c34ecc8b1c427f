# round 4: kernel trace of the training step (steady state = the last 370 ms) + where the GPU idles
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
export VPHO_WGRAD_STREAM=0          # one stream: the trace's durations are exclusive (the timed bench line overlaps the weight gradients)
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/r04_train_prof -o t -- python3 $R/train.py --steps 5 --warmup 2 > $O/r04_train_prof.json 2> $O/r04_train_prof.err && echo train-prof-done
cd $R && DB=$(ls $O/r04_train_prof/*/*.db $O/r04_train_prof/*.db 2>/dev/null | head -1)
python3 scripts/rocpd_stats.py $DB 40 --last-ms 370 > $O/r04_train_stats.txt; python3 scripts/rocpd_gaps.py $DB >> $O/r04_train_stats.txt 2>&1
rm -rf $O/r04_train_prof; head -50 $O/r04_train_stats.txt | cut -c1-160
