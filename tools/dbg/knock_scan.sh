for k in ${KS:-0 64 66 68 80 70}; do echo "KNOCK=$k"; MIVI_KNOCK=$k tools/dbg/iso20_prof.sh 12 | grep -E "prod|vjp" | tail -6 | awk '{print $1, $5}' | tr '\n' ' '; echo; done
