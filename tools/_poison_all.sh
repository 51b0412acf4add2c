for c in "DistMult 1" "ComplEx 1" "ComplEx 0" "RotatE 0" "TransE 0" "pRotatE 0"; do
  echo "== $c"; timeout 300 python tools/_poison_check.py $c 2>&1 | grep -v amdgpu | grep -i "DIFF\|fault\|NON-FINITE\|Error\|SAME" | head -8
done
