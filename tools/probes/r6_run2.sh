set -x
SCRIPT=train_resnet_prof.py STEPS=8 PERIODS=5 MAXL=1600 bash tools/probes/seq_any.sh > gpurun_out/r6_seq_resnet_train.txt 2>&1
SCRIPT=train_chain_prof.py PERIODS=4 MAXL=600 bash tools/probes/seq_any.sh > gpurun_out/r6_seq_alexnet_train.txt 2>&1
bash tools/probes/c3_pmc.sh > gpurun_out/r6_c3_pmc.log 2>&1
tail -3 gpurun_out/r6_seq_resnet_train.txt gpurun_out/r6_seq_alexnet_train.txt
