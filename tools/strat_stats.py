import sys
sys.path.insert(0,'.')
import numpy as np, torch
import __graft_entry__, bench
pkg = __graft_entry__.load_package()
size = 4096
dev = torch.device("cuda", 0)
frame = bench.frame_rows_on_device(torch, size, 0, size, 0, dev)
enc = pkg.Encoder(0)
enc.set_device_image([frame[c].data_ptr() for c in range(3)], size * 4, size, size, keepalive=frame)
enc.enqueue(1.0, 0); enc.synchronize()
fr = enc.fetch_raw()
st = np.ctypeslib.as_array(fr.ac_strategy, shape=(fr.xsize_blocks * fr.ysize_blocks,)).copy()
vals, cnt = np.unique(st, return_counts=True)
print('strategy byte (code << 1 | first): share of blocks', dict(zip(vals.tolist(), (cnt/len(st)).round(4).tolist())))
g = st.reshape(fr.ysize_blocks, fr.xsize_blocks)
# per wave of tile_kernel (8 octets = one block row of a tile... candidates are cells): share of 8x8-tile rows with no two-block transform at all
t = (g >> 1).reshape(fr.ysize_blocks // 8, 8, fr.xsize_blocks // 8, 8)
print('tiles without any two-block transform: %.3f' % float(((t != 0).sum(axis=(1, 3)) == 0).mean()))
