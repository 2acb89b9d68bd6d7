# On the GPU box: the whole GPU suite under every remaining A/B environment switch (each selects an in-library reference route;
# tests/test_gpu_ab_switches.py covers each switch at one shape inside the default suite, this sweeps the whole suite).
for e in MIVI_NO_FUSED_LOOP=1 MIVI_LR_F32_XTR=1 MIVI_LR_F32_LOGITS=1 MIVI_LOGREG_GENERIC=1 MIVI_LOGREG_MFMA=1 \
         MIVI_NO_FUSED_UPDATE=1 MIVI_STL_VALU=1 MIVI_STL_GEN1=1 MIVI_F64_VALU=1 MIVI_FR_GEN1=1 MIVI_FR_F32MFMA=1 MIVI_STEIN_GEN1=1 MIVI_VJP_TILE=64; do
  echo "== $e"; env $e timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
done
