for pc in 2 1; do echo "PER_CU $pc"; QT_C3_PER_CU=$pc bash tools/probes/c4_kt.sh 2>&1 | grep -i "code_conv3x3"; done
