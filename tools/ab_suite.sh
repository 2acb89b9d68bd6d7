# On the GPU box: the whole GPU suite under every A/B environment switch (each selects a reference / older kernel route).
for e in MIVI_NO_FUSED_LOOP=1 MIVI_LR_F32_XTR=1 MIVI_LR_F32_LOGITS=1 MIVI_LR_GEN1=1 MIVI_LR_XTR_NARROW=1 MIVI_LOGREG_GENERIC=1 MIVI_LOGREG_MFMA=1 \
         MIVI_NO_FUSED_UPDATE=1 MIVI_STL_VALU=1 MIVI_F64_VALU=1 MIVI_FR_GEN1=1 MIVI_FR_F32MFMA=1 MIVI_STEIN_GEN1=1 MIVI_PROD64=0 MIVI_VJP_TILE=64 \
         MIVI_LDS_SPLITK=1; do
  echo "== $e"; env $e timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
done
