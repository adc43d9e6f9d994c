#!/usr/bin/env python3
"""Crude BC7 encoder (modes 4, 5, 6; one subset each) over a procedural RGBA image: a corpus with spatially correlated
endpoints and structured indices, to judge BC7 stream layouts on more than the 4096 blocks of the reference's one BC7
test texture (docs/BC7_FORMAT.md section 4).  Test tooling only; not a product path.

    python tools/bc7_synth.py 1024 /tmp/bc7_synth_1024.bin      # 1024 x 1024 px = 65 536 blocks
"""
import numpy as np
def smooth_noise(h,w,rng,octaves=5):
    img=np.zeros((h,w))
    for o in range(octaves):
        s=2**(o+2)
        g=rng.random((s+1,s+1))
        ys=np.linspace(0,s,h,endpoint=False); xs=np.linspace(0,s,w,endpoint=False)
        y0=ys.astype(int); x0=xs.astype(int); fy=(ys-y0)[:,None]; fx=(xs-x0)[None,:]
        fy=fy*fy*(3-2*fy); fx=fx*fx*(3-2*fx)
        a=g[y0][:,x0]; b=g[y0][:,x0+1]; c=g[y0+1][:,x0]; d=g[y0+1][:,x0+1]
        img+= (a*(1-fx)+b*fx)*(1-fy)+(c*(1-fx)+d*fx)*fy
        img*=1.0
        if o<octaves-1: img*=1.0
    img-=img.min(); img/=img.max(); return img
def make_image(n=1024,seed=7):
    rng=np.random.default_rng(seed)
    chans=[smooth_noise(n,n,rng,6) for _ in range(3)]
    base=smooth_noise(n,n,rng,3)
    rgb=np.stack([np.clip(0.6*base+0.5*c-0.05,0,1) for c in chans],-1)
    rgb+=rng.normal(0,0.01,rgb.shape)   # sensor-like noise
    a=smooth_noise(n,n,rng,4)
    alpha=np.clip((a-0.45)*4,0,1)       # large opaque and transparent regions with soft borders
    img=np.concatenate([np.clip(rgb,0,1),alpha[...,None]],-1)
    return (img*255+0.5).astype(np.uint8)
W4=np.array([0,4,9,13,17,21,26,30,34,38,43,47,51,55,60,64]); W3=np.array([0,9,18,27,37,46,55,64]); W2=np.array([0,21,43,64])
def quant(v,bits):  # 8-bit -> bits
    return np.clip((v.astype(np.int32)*((1<<bits)-1)+127)//255,0,(1<<bits)-1)
def expand(q,bits):
    v=q<<(8-bits); return v|(v>>bits)
def project(px,e0,e1,weights):
    # px: (16,C) ints, e0,e1: (C,) ints; index of nearest interpolated colour
    pal=((64-weights)[:,None]*e0[None,:]+weights[:,None]*e1[None,:]+32)>>6
    d=((px[:,None,:].astype(np.int32)-pal[None,:,:])**2).sum(-1)
    return d.argmin(1)
class BitW:
    def __init__(s): s.v=0; s.n=0
    def put(s,val,w): s.v|=(int(val)&((1<<w)-1))<<s.n; s.n+=w
def enc6(px):
    lo=px.min(0).astype(np.int32); hi=px.max(0).astype(np.int32)
    # p-bit: parity that fits best (use lsb of mean of lo / hi)
    p0=int(lo.sum()&1); p1=int(hi.sum()&1)
    q0=np.clip((lo-p0+1)>>1,0,127); q1=np.clip((hi-p1+1)>>1,0,127)
    e0=(q0<<1)|p0; e1=(q1<<1)|p1
    idx=project(px,e0,e1,W4)
    if idx[0]>=8: q0,q1,p0,p1=q1,q0,p1,p0; idx=15-idx
    b=BitW(); b.put(1<<6,7)
    for c in range(4): b.put(q0[c],7); b.put(q1[c],7)
    b.put(p0,1); b.put(p1,1)
    b.put(idx[0],3)
    for i in range(1,16): b.put(idx[i],4)
    assert b.n==128; return b.v
def enc5(px):
    lo=px.min(0).astype(np.int32); hi=px.max(0).astype(np.int32)
    qc0=quant(lo[:3],7); qc1=quant(hi[:3],7); a0=int(lo[3]); a1=int(hi[3])
    e0=expand(qc0,7); e1=expand(qc1,7)
    ci=project(px[:,:3],e0,e1,W2)
    if ci[0]>=2: qc0,qc1=qc1,qc0; ci=3-ci
    ai=project(px[:,3:],np.array([a0]),np.array([a1]),W2)
    if ai[0]>=2: a0,a1=a1,a0; ai=3-ai
    b=BitW(); b.put(1<<5,6); b.put(0,2)
    for c in range(3): b.put(qc0[c],7); b.put(qc1[c],7)
    b.put(a0,8); b.put(a1,8)
    b.put(ci[0],1)
    for i in range(1,16): b.put(ci[i],2)
    b.put(ai[0],1)
    for i in range(1,16): b.put(ai[i],2)
    assert b.n==128; return b.v
def enc4(px):
    lo=px.min(0).astype(np.int32); hi=px.max(0).astype(np.int32)
    qc0=quant(lo[:3],5); qc1=quant(hi[:3],5); a0=quant(lo[3:],6)[0]; a1=quant(hi[3:],6)[0]
    e0=expand(qc0,5); e1=expand(qc1,5)
    ci=project(px[:,:3],e0,e1,W2)
    if ci[0]>=2: qc0,qc1=qc1,qc0; ci=3-ci
    ai=project(px[:,3:],expand(np.array([a0]),6),expand(np.array([a1]),6),W3)
    if ai[0]>=4: a0,a1=a1,a0; ai=7-ai
    b=BitW(); b.put(1<<4,5); b.put(0,2); b.put(0,1)
    for c in range(3): b.put(qc0[c],5); b.put(qc1[c],5)
    b.put(a0,6); b.put(a1,6)
    b.put(ci[0],1)
    for i in range(1,16): b.put(ci[i],2)
    b.put(ai[0],2)
    for i in range(1,16): b.put(ai[i],3)
    assert b.n==128; return b.v
def encode(img,seed=3):
    rng=np.random.default_rng(seed)
    h,w,_=img.shape; out=bytearray()
    for by in range(0,h,4):
        for bx in range(0,w,4):
            px=img[by:by+4,bx:bx+4].reshape(16,4)
            arange=int(px[:,3].max())-int(px[:,3].min())
            if arange==0 and px[0,3]==255: v=enc6(px)
            elif arange==0: v=enc6(px) if rng.random()<0.5 else enc5(px)
            else: v=enc5(px) if rng.random()<0.7 else enc4(px)
            out+=v.to_bytes(16,'little')
    return bytes(out)
if __name__=='__main__':
    import sys
    n=int(sys.argv[1]) if len(sys.argv)>1 else 512
    out=sys.argv[2] if len(sys.argv)>2 else f'/tmp/bc7_synth_{n}.bin'
    img=make_image(n)
    data=encode(img)
    open(out,'wb').write(data)
    b0=np.frombuffer(data,dtype=np.uint8)[::16]
    print(len(data)//16,'blocks', np.bincount([ ((int(v)|0x100)&-(int(v)|0x100)).bit_length()-1 for v in b0],minlength=9))
