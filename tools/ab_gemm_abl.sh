# diagnostic: where a gemm256 tile spends its time (ablation builds under gpurun_out/, results wrong by construction)
for lib in "" abl/libvla_abl1.so abl/libvla_abl3.so; do
  echo "== lib ${lib:-product}"
  VLA_NATIVE_LIB=$lib SHAPES="gate_up(swiglu),llm down,vit fc1,vit qkv,vit fc2,square 4096" timeout -k 10 200 python tools/bench_gemm256.py 2>/dev/null | sed 's/| vendor.*//'
done
