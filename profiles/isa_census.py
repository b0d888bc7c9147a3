#!/usr/bin/env python3
"""profiles/isa_census.py -- what the gfx950 code objects of the built library hold: run after
   for n in expann_hip expann_graph expann_sharded; do
     llvm-objcopy --dump-section .hip_fatbin=/tmp/isa/$n.fat expann_amd/csrc/_obj/$n.o
     clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/isa/$n.fat --output=/tmp/isa/$n.co
     llvm-objdump -d /tmp/isa/$n.co > /tmp/isa/$n.s
   done
(tools under /opt/rocm/lib/llvm/bin).  Counts of the instructions that say how the kernels are written -- and of
scratch_ instructions, which must be 0 (profiles/r03_isa_census.txt)."""
import re, sys, collections, subprocess
B='/opt/rocm/lib/llvm/bin/'
out=[]
tot=collections.Counter()
for n in ('expann_hip','expann_graph','expann_sharded'):
    try:
        txt=open('/tmp/isa/%s.s'%n).read()
    except Exception:
        continue
    if not txt: 
        out.append("%s: no device code of its own (its kernels are expann_hip's)"%n); continue
    c=collections.Counter(re.findall(r'\b(v_mfma_[a-z0-9_]+|scratch_[a-z0-9_]+|global_load_lds_[a-z0-9]+|ds_read_b128|ds_read_b64_tr_b16|v_dot4[a-z0-9_]*)\b', txt))
    dpp=len(re.findall(r'_dpp|row_ror|row_shr|quad_perm', txt))
    kern=len(re.findall(r'^[0-9a-f]+ <[^>]+>:', txt, re.M))
    out.append("%s: %d functions; "%(n,kern)+", ".join("%s %d"%(k,v) for k,v in sorted(c.items()))+", DPP-modified instructions %d"%dpp)
    tot.update(c)
print("\n".join(out))
print("scratch instructions in the whole library:", sum(v for k,v in tot.items() if k.startswith('scratch_')))
