for cfg in "128 128 16 0 1" "128 128 8 0 1" "64 64 32 0 1" "64 64 64 0 1" "32 32 64 0 1" "64 128 16 1 1" "128 64 32 2 1" "128 128 16 0 2" "64 64 32 0 2" "32 64 32 1 1"; do set -- $cfg
 for m in 0 1; do NGAN_MID_F32=$m timeout -k 10 120 python tools/conv_micro.py --op fwd --B 16 --H $3 --W $3 --K $1 --N $2 --res $4 --epi $5 --prec 0 2>&1 | grep -v amdgpu.ids | sed "s/^/mid=$m /"; done; done
