for m in grp_1_75 grp_1_150 grp_4_75 grp_16_150 grp_1_20; do timeout -k 10 200 python tools/mode_probe.py $m 2>&1 | grep -E "mode|groups"; done
