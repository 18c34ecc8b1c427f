# round 4 (late): training-step bench line + kernel trace after the packed-layout optimiser state, the input-gradient weight cache and the weight-gradient stream
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 300 python train.py --steps 10 --warmup 3 > $O/r04_train_step.json 2> $O/r04_train_step.err && echo train-done &&
bash scripts/r04_train_prof.sh
