cd /root/repo
python -m pytest tests/test_gpu_skinny_lds.py tests/test_gpu_llm.py -x -q 2>&1 | tail -3
for b in 64 128; do
echo "== new B=$b"; python tools/microbench.py --batch $b --only dec_ 2>&1 | grep -E "^decode x-through-LDS" 
done
