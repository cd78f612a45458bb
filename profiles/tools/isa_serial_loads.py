"""Static scan of a translation unit's gfx950 ISA for serialised memory chains: waits for ALL outstanding vector-memory operations (s_waitcnt vmcnt(0))
that follow at most two loads since the previous wait -- a load, a full drain, the next load: one round trip each (how k_egnn_edge_bwd's global-vector
row dots and k_node_update8's piece gather were found).  python profiles/tools/isa_serial_loads.py keypoint-diffusion_amd/csrc/egnn_kernels.hip [more.hip]"""
import re, subprocess, sys, os, tempfile

for src in sys.argv[1:]:
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + '.s')
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '--cuda-device-only', '-S', src, '-o', out], check=True,
                   stderr=subprocess.DEVNULL)
    text = open(out).read()
    print(src)
    for m in re.finditer(r'\n(_Z\w+):[^\n]*\n', text):
        name = m.group(1)
        end = text.find('s_endpgm', m.end())
        body = text[m.end():end].splitlines()
        if len(body) < 50:
            continue
        loads = 0
        waits = []
        for k, l in enumerate(body):
            if 'global_load' in l or 'scratch_load' in l or 'buffer_load' in l:
                loads += 1
            w = re.search(r's_waitcnt.*vmcnt\((\d+)\)', l)
            if w:
                waits.append((int(w.group(1)), loads))
                loads = 0
        sus = sum(1 for c, n in waits if c == 0 and 1 <= n <= 2)
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        print(f'  {dem[:90]:90s} lines {len(body):6d}  vmcnt waits {len(waits):4d}  drain-after-1-2-loads {sus:4d}')
