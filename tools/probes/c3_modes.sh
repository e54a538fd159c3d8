for m in 0 2 3 4; do echo "MODE $m"; QT_C3_MODE=$m bash tools/probes/c4_kt.sh 2>&1 | grep -i "code_conv3x3"; done
