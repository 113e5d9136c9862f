# key_chars sweep: bash tests/tools/kc2.sh words 29 "6 8 9 10 11 12"
corpus=$1; logn=$2
for k in $3; do echo -n "key_chars=$k: "; PSS_KEY_CHARS=$k python tests/tools/sa_perf.py $corpus $logn 3 2>&1 | grep -E "rep 2" | sed -E "s/.*dev ([0-9.]+) ms.*initial_passes.: ([0-9]+).*'rounds.: ([0-9]+).*round_passes.: ([0-9]+).*sum_active.: ([0-9]+).*big_elems.: ([0-9]+).*mid_elems.: ([0-9]+).*/\1 ms passes=\2 rounds=\3 round_passes=\4 sum_active=\5 big=\6 mid=\7/"; done
