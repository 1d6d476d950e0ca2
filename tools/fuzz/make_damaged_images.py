import sys; import os; R=os.path.join(os.path.dirname(os.path.abspath(__file__)),"..",".."); sys.path.insert(0,os.path.join(R,"tests")); sys.path.insert(0,R)
import numpy as np, pathlib
from PIL import Image
import test_image_textures as t
rng=np.random.default_rng(int(sys.argv[1]))
d=pathlib.Path(sys.argv[2]); d.mkdir(parents=True,exist_ok=True)
rgb=t._jpeg_test_image(37,51)
seeds={}
Image.fromarray(rgb,"RGB").save(d/"b.jpg",quality=85,subsampling=2,restart_marker_blocks=3); seeds["base.jpg"]=open(d/"b.jpg","rb").read()
Image.fromarray(rgb,"RGB").save(d/"b2.jpg",quality=60,subsampling=1); seeds["b422.jpg"]=open(d/"b2.jpg","rb").read()
Image.fromarray(rgb,"RGB").save(d/"p.jpg",quality=85,subsampling=2,progressive=True); seeds["prog.jpg"]=open(d/"p.jpg","rb").read()
Image.fromarray(rgb[...,0],"L").save(d/"g.jpg",quality=85,progressive=True); seeds["gprog.jpg"]=open(d/"g.jpg","rb").read()
a=(rgb[...,:3]/255.0).astype(np.float16)
t.write_exr(d/"z.exr",{"R":(a[...,0],"half"),"G":(a[...,1].astype(np.float32),"float"),"B":(a[...,2],"half")},4); seeds["piz.exr"]=open(d/"z.exr","rb").read()
t.write_exr(d/"x.exr",{"R":(a[...,0],"half"),"G":(a[...,1],"half"),"B":(a[...,2],"half")},5); seeds["pxr.exr"]=open(d/"x.exr","rb").read()
t.write_exr(d/"y.exr",{"R":(a[...,0],"half"),"G":(a[...,1],"half"),"B":(a[...,2],"half")},3); seeds["zip.exr"]=open(d/"y.exr","rb").read()
t.write_exr(d/"t.exr",{"R":(a[...,0],"half"),"G":(a[...,1].astype(np.float32),"float"),"B":(a[...,2],"half")},4,version=2|0x200,tiles=(32,16,1)); seeds["tiledpiz.exr"]=open(d/"t.exr","rb").read()
t.write_exr(d/"u.exr",{"R":(a[...,0],"half"),"G":(a[...,1],"half"),"B":(a[...,2],"half")},3,version=2|0x200,tiles=(16,16,0)); seeds["tiledzip.exr"]=open(d/"u.exr","rb").read()
def exr_table_at(b):
    """Byte position of an EXR file's offset table: behind the header's attribute list (name\\0 type\\0 size value ... \\0)."""
    pos=8
    while b[pos]!=0:
        pos=b.index(0,pos)+1; pos=b.index(0,pos)+1
        pos+=4+int.from_bytes(b[pos:pos+4],"little",signed=True)
    return pos+1
def jpeg_sof_at(b):
    pos=2
    while pos+4<len(b):
        if b[pos]==0xff and b[pos+1] in (0xc0,0xc1,0xc2): return pos+5      # height, width: two big-endian u16 each
        pos+=2+((b[pos+2]<<8)|b[pos+3])
    return None
EXTREME=[2**64-1,2**64-16,2**64-20,2**64-8,2**63,2**63-1,2**32,2**32-1,2**31]
n=int(sys.argv[3])
# directed cases first: values a random byte flip never produces -- 64-bit block offsets whose sum with a header size wraps, offsets at
# the very end of the file, and frame headers that claim the largest image the format can describe
k=0
for name,seed in seeds.items():
    if name.endswith(".exr"):
        at=exr_table_at(seed)
        for v in EXTREME+[len(seed),len(seed)-1,len(seed)-7,len(seed)-19]:
            for entry in (0,1):
                data=bytearray(seed); data[at+8*entry:at+8*entry+8]=int(v).to_bytes(8,"little")
                open(d/("f_dir%04d_%s"%(k,name)),"wb").write(data); k+=1
    else:
        at=jpeg_sof_at(seed)
        for hw in ((65535,65535),(65535,8),(8,65535),(0,0),(1,65535)):
            data=bytearray(seed); data[at:at+4]=hw[0].to_bytes(2,"big")+hw[1].to_bytes(2,"big")
            open(d/("f_dir%04d_%s"%(k,name)),"wb").write(data); k+=1
        cut=seed.index(b"\xff\xda")
        ln=(seed[cut+2]<<8)|seed[cut+3]
        open(d/("f_dir%04d_%s"%(k,name)),"wb").write(seed[:cut+2+ln]); k+=1      # the scan header ends the file
for it in range(n):
    name=list(seeds)[it%len(seeds)]
    data=bytearray(seeds[name])
    mode=rng.integers(0,4)
    if mode==0:
        for _ in range(rng.integers(1,6)): data[rng.integers(0,len(data))]=rng.integers(0,256)
    elif mode==1:
        data=data[:rng.integers(4,len(data))]
    elif mode==2:
        i=rng.integers(0,len(data)); data[i:i+rng.integers(1,40)]=bytes(rng.integers(0,256,rng.integers(0,40),dtype=np.uint8))
    else:   # header-area mutation
        for _ in range(rng.integers(1,4)): data[rng.integers(0,min(700,len(data)))]=rng.integers(0,256)
    open(d/("f%05d_%s"%(it,name)),"wb").write(data)
