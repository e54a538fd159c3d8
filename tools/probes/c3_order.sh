for o in xcd plain xcd plain; do echo "ORDER $o"; if [ $o = plain ]; then export QT_C3_PLAIN_ORDER=1; else unset QT_C3_PLAIN_ORDER; fi; bash tools/probes/c4_kt.sh 2>&1 | grep -i "code_conv3x3"; done
