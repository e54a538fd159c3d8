# training-path GPU tests + step times + launch counts after the round-6 launch diet
python -m pytest tests/test_gpu_r2.py tests/test_gpu_r3.py tests/test_gpu_r4.py tests/test_gpu_r5.py tests/test_gpu_r5d.py tests/test_gpu_r3b.py -q 2>&1 | tail -12
python tools/probes/train_resnet_time.py 2>&1 | tail -9 | head -5
python tools/probes/train_chain_time.py 2>&1 | tail -2
SCRIPT=train_resnet_prof.py STEPS=8 PERIODS=5 MAXL=1600 bash tools/probes/seq_any.sh > gpurun_out/r6c_seq_resnet_train.txt 2>&1
tail -1 gpurun_out/r6c_seq_resnet_train.txt
python tools/probes/train_glue_sources.py > gpurun_out/r6c_glue_resnet.txt 2>&1; MODEL=alexnet python tools/probes/train_glue_sources.py > gpurun_out/r6c_glue_alexnet.txt 2>&1
SCRIPT=train_chain_prof.py PERIODS=4 MAXL=600 bash tools/probes/seq_any.sh > gpurun_out/r6c_seq_alexnet_train.txt 2>&1
tail -1 gpurun_out/r6c_seq_alexnet_train.txt
