for e in MIVI_NO_FUSED_LOOP MIVI_LR_F32_XTR MIVI_LR_F32_LOGITS MIVI_LR_GEN1 MIVI_LR_XTR_NARROW MIVI_LOGREG_GENERIC MIVI_NO_FUSED_UPDATE MIVI_STL_VALU MIVI_F64_VALU; do
  echo "== $e"; env $e=1 timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -1
done
