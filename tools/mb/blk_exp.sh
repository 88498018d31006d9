# A/B of the activation layouts on the Winograd kernels of the product build (timing; MB_LAYOUT = x, y digits, 1 = KFN_LAYOUT_C16)
for rep in 1 2; do
for lay in 00 10 11 01; do
  echo "== layout $lay rep $rep"
  MB_LAYOUT=$lay MB_BATCH=20 MB_F43_FORM=3 MB_FUSED_ONLY=1 MB_LAYERS=conv1b,conv2b,conv3b,conv4b,conv5,conv6 python tools/mb_wino.py 2>&1 | grep -o "^conv[0-9a-z]*\|F(4x4,3x3) [0-9.]* ms" | paste - -
  MB_LAYOUT=$lay MB_BATCH=20 MB_S2_FORM=5 python tools/mb_s2.py 2>&1 | grep -o "^conv[0-9a-z]*\|polyphase [0-9.]* ms" | paste - -
done; done
