for a in 0 1 2 3 4 5 6; do echo "== ABL $a"; OCS_FCS_ABL=$a timeout -k 10 100 scripts/prof_bl4.sh abl$a "8192" on 2>&1 | grep -E "k_backward" | cut -d, -f1-4 | cut -c1-120; done
