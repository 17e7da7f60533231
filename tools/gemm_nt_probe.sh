for shape in "58051 128 343 64" "219698 32 343 16"; do
for e in "PCC_NT=0" "PCC_NT=1" "PCC_NT=1 PCC_DBG=1" "PCC_NT=1 PCC_DBG=2" "PCC_NT=1 PCC_DBG=4" "PCC_NT=1 PCC_DBG=6"; do
  echo "== $shape $e"; env $e python tools/gemm_h2_probe.py $shape 6 2>/dev/null | grep gemm_h2
done; done
