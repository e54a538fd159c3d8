for m in ${MODES:-0 2 5}; do echo "MODE $m"; QT_C3_MODE=$m bash tools/probes/c4_kt.sh 2>&1 | grep -i "code_conv3x3"; done
