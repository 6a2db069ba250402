for rep in 1 2; do for v in base dense; do
 if [ $v = base ]; then lib=""; else lib=$PWD/flydog_sdr_gps_amd/libkiwigpu_$v.so; fi
 echo "== $v"; KIWIGPU_LIBRARY=$lib python3 tools/time_e1b.py 32 2>&1 | tail -1
done; done
