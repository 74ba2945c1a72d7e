"""Host-side model of the LDS bank conflicts of conv3x3_f16dma_kernel's A-fragment reads (ds_read_b128: four groups of 16 lanes, 64 banks of 4 B;
a group is conflict-free iff its 16 addresses fall into 16 different 16-byte bank quads -- MI355X_MICROARCH.md, LDS table).  The halo is stored
pixel-major, 128 B per pixel, chunk slot = chunk ^ ((hp >> 1) & 7): the bank quad of a read is a bijection of hp mod 16, so a group is
conflict-free iff its lanes' halo-pixel indices are distinct mod 16.  64- and 32-column images (32 lanes = one image row): 0 extra cycles;
16-column images (two rows per 32 lanes, row pitch 18 = 2 mod 16): +1 cycle in EVERY group (4 -> 5); 8-column images (four rows, pitch 10): +2
(4 -> 6).  Matches the SQ counters of the closing code (profiles/r4_*_fp16_sq_counters.json: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.25 - 0.29
on the <16, .> instantiations, 0.43 on <8, .>, 0.00 on <32, .> / <64, .>).  No XOR of the chunk bits with a per-row constant removes it (searched
below); a swizzle on the COLUMN of the halo pixel, chunk ^ ((v >> 1) & 7) with v = x (16 columns) / v = x + 8 (row & 1) (8 columns), does -- see
docs/HISTORY.md section E.21.  python tools/probes/halo_bank_conflicts.py"""
import itertools, sys
G0=[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27]; G1=[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]
def geo(W):
    NIMG = 1 if W*W>=256 else 256//(W*W)
    TH = 256//(W*NIMG); WP=W+2; HP=TH+2
    return NIMG,TH,WP,HP
def conflicts(W, X, WP_override=None):
    NIMG,TH,WP,HP = geo(W)
    if WP_override: WP=WP_override
    extra=0; total=0
    for wr in range(4):
        for i in range(2):
            for tt in range(9):
                for g in (G0,G1):
                    quads={}
                    for lane in g:
                        m = wr*64 + i*32 + lane
                        sl = m//(TH*W); rem = m - sl*TH*W; r = rem//W; c = rem - r*W
                        hr = r + tt//3; hc = c + tt%3
                        hp = (sl*HP + hr)*WP + hc
                        chunk = ((hp>>1)&7) ^ X(sl*HP+hr, hp)
                        a16 = hp*8 + chunk
                        quads.setdefault(a16%16,set()).add(a16)
                    extra += max(len(v) for v in quads.values())-1; total += 1
    return extra, total
for W in (64,32,16,8):
    print(W, 'current (extra cycles, groups)', conflicts(W, lambda row,hp: 0), flush=True)
for W,mask in ((16,1),(16,3),(8,1),(8,3)):
    best=None
    for tab in itertools.product(range(8), repeat=mask+1):
        if tab[0]!=0: continue
        e,t = conflicts(W, lambda row,hp,tab=tab,mask=mask: tab[row&mask])
        if best is None or e<best[0]: best=(e,tab)
        if e==0: break
    print(W, 'row-xor table mask', mask, 'best', best, flush=True)
# the column-based swizzle of HISTORY E.21: conflict-free on every image width
def by_column(W):
    NIMG,TH,WP,HP = geo(W)
    def run():
        extra=0; total=0
        for wr in range(4):
            for i in range(2):
                for tt in range(9):
                    for g in (G0,G1):
                        quads={}
                        for lane in g:
                            m = wr*64 + i*32 + lane
                            sl = m//(TH*W); rem = m - sl*TH*W; r = rem//W; c = rem - r*W
                            hr = r + tt//3; hc = c + tt%3
                            hp = (sl*HP + hr)*WP + hc
                            v = hc + (8*((sl*HP+hr)&1) if W == 8 else 0) if W <= 16 else hp
                            a16 = hp*8 + ((v>>1)&7)
                            quads.setdefault(a16%16,set()).add(a16)
                        extra += max(len(v_) for v_ in quads.values())-1; total += 1
        return extra, total
    return run()
for W in (64,32,16,8):
    print(W, 'column-based swizzle', by_column(W), flush=True)
# alternative: change the halo row pitch
for W in (16,8):
    for WP in range(W+2, W+2+16):
        print(W, 'pitch', WP, conflicts(W, lambda row,hp:0, WP), flush=True)
