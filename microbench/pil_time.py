import time, numpy as np
from PIL import Image
rng=np.random.default_rng(0)
a=rng.integers(0,256,(4096,4096,3),dtype=np.uint8)
im=Image.fromarray(a,mode="YCbCr")
def best(f,n=5):
    ts=[]
    for _ in range(n):
        t0=time.perf_counter(); r=f(); ts.append(time.perf_counter()-t0); del r
    return min(ts)*1e3
print("image.split(): %.2f ms"%best(lambda: im.split()))
print("split + asarray per band: %.2f ms"%best(lambda: [np.asarray(b) for b in im.split()]))
print("np.asarray(image) (H,W,3): %.2f ms"%best(lambda: np.asarray(im)))
print("np.ascontiguousarray(np.asarray(image)): %.2f ms"%best(lambda: np.ascontiguousarray(np.asarray(im))))
x=np.asarray(im)
print("numpy deinterleave x[...,k].copy() x3: %.2f ms"%best(lambda: [np.ascontiguousarray(x[...,k]) for k in range(3)]))
rgb=Image.fromarray(a,mode="RGB")
print("RGB -> YCbCr convert: %.2f ms"%best(lambda: rgb.convert("YCbCr")))
print("Image.fromarray(packed, YCbCr): %.2f ms"%best(lambda: Image.fromarray(a,mode="YCbCr")))
print("YCbCr -> RGB convert: %.2f ms"%best(lambda: im.convert("RGB")))
try:
    print('im.tobytes("raw", "YCbCrX"): %.2f ms' % best(lambda: im.tobytes("raw", "YCbCrX")))
except Exception as e:
    print("tobytes YCbCrX:", e)
try:
    print('im.tobytes("raw", "YCbCr"): %.2f ms' % best(lambda: im.tobytes("raw", "YCbCr")))
except Exception as e:
    print("tobytes YCbCr:", e)
b4 = np.zeros((4096, 4096, 4), np.uint8)
b4[..., :3] = a
buf4 = b4.tobytes()
for rawmode in ("YCbCrX", "YCbCr;L"):
    try:
        print('Image.frombuffer("YCbCr", ..., "raw", %r): %.2f ms' % (rawmode, best(lambda: Image.frombuffer("YCbCr", (4096, 4096), buf4, "raw", rawmode, 0, 1))))
    except Exception as e:
        print("frombuffer", rawmode, e)
print('Image.frombuffer("YCbCr", ..., "raw", "YCbCr") 3 bytes: %.2f ms' % best(lambda: Image.frombuffer("YCbCr", (4096, 4096), a.tobytes(), "raw", "YCbCr", 0, 1)))
abytes = a.tobytes()
print('Image.frombuffer 3 bytes, bytes ready: %.2f ms' % best(lambda: Image.frombuffer("YCbCr", (4096, 4096), abytes, "raw", "YCbCr", 0, 1)))
y = a[..., 0].copy()
print('Image.fromarray(L band): %.2f ms' % best(lambda: Image.fromarray(y, mode="L")))
ybands = [Image.fromarray(np.ascontiguousarray(a[..., k]), mode="L") for k in range(3)]
print('Image.merge("YCbCr", 3 L bands): %.2f ms' % best(lambda: Image.merge("YCbCr", ybands)))
print('Image.frombuffer("L") x3 + merge: %.2f ms' % best(lambda: Image.merge("YCbCr", [Image.frombuffer("L", (4096, 4096), np.ascontiguousarray(a[..., k]), "raw", "L", 0, 1) for k in range(3)])))
