for v in base skip; do echo "== $v"; AHIP_LIB=/root/repo/archive_amd/lib/var_$v.so timeout -k 5 120 python tools/deflate_quick.py 2>&1 | grep level; done
