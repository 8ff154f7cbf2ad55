#!/usr/bin/env python3
"""CPU model (no GPU): size of the image patch that a BLOCK of voxels touches per view (bounding box of the 2x2 tap
origins), against the number of tap pixels the block gathers - how much a kernel that stages the patch in LDS could
save in bytes through the texture-address path.  (The kernel was built in round 2 and lost: DESIGN.md 7b.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from selfpose3d_amd import synthetic as syn
def setup(V, cube, space, center, w=240, h=128, img=(960,512)):
    X, Y, Z = cube
    cams = syn.ring_cameras(V)
    gx = np.linspace(-space[0]/2, space[0]/2, X) + center[0]
    gy = np.linspace(-space[1]/2, space[1]/2, Y) + center[1]
    gz = np.linspace(-space[2]/2, space[2]/2, Z) + center[2]
    P = np.stack(np.meshgrid(gx, gy, gz, indexing="ij"), -1).reshape(-1, 3)
    a = img[0] / (200 * syn.get_scale(syn.ORIG_IMAGE, img)[0])
    out=[]
    for c in range(V):
        px = syn._project_f64(P, cams[c])
        bound = (px[:, 0] >= 0) & (px[:, 1] >= 0) & (px[:, 0] < 1920) & (px[:, 1] < 1080)
        px = np.nan_to_num(np.clip(px, -1, 1920))
        q = (px - np.array([960, 540])) * a + np.array([img[0] / 2, img[1] / 2])
        ix, iy = q[:, 0] * w / img[0], q[:, 1] * h / img[1]
        x0 = np.clip(np.floor(ix).astype(int), 0, w - 2); y0 = np.clip(np.floor(iy).astype(int), 0, h - 2)
        out.append((x0,y0,bound))
    return out
def groups(cube, shp):
    X, Y, Z = cube; tx, ty, tz = shp
    idx = np.arange(X*Y*Z).reshape(X, Y, Z)
    return idx.reshape(X//tx, tx, Y//ty, ty, Z//tz, tz).transpose(0,2,4,1,3,5).reshape(-1, tx*ty*tz)
def run(name, V, cube, space, center, shapes, cap):
    views = setup(V, cube, space, center)
    for shp in shapes:
        t = groups(cube, shp)
        areas=[]; taps=0; fit=0; tot=0; fit_px=0; fit_taps=0
        for (x0,y0,b) in views:
            X0=x0[t]; Y0=y0[t]; B=b[t]
            any_=B.any(1)
            big=10**6
            xmin=np.where(B,X0,big).min(1); xmax=np.where(B,X0,-1).max(1)
            ymin=np.where(B,Y0,big).min(1); ymax=np.where(B,Y0,-1).max(1)
            W=(xmax-xmin+2); H=(ymax-ymin+2)
            A=(W*H)[any_]
            nt=B.sum(1)[any_]*4
            areas.append(A); tot+=len(A); ok=A<=cap; fit+=ok.sum(); fit_px+=A[ok].sum(); fit_taps+=nt[ok].sum(); taps+=nt.sum()
        A=np.concatenate(areas)
        print(f"{name} block {shp}: median patch {np.median(A):.0f} px, p90 {np.percentile(A,90):.0f}, fit(<= {cap}) {fit/tot:.2f}; taps in fitting blocks {fit_taps/taps:.2f}; patch px / tap px there {fit_px/fit_taps:.2f}")
run("fine", 5, (64,64,64), syn.FINE_GRID_SIZE, (0.,-500.,800.), [(8,8,8),(4,4,32),(8,8,4),(4,4,16),(8,8,16)], 640)
run("fine(off-centre)", 5, (64,64,64), syn.FINE_GRID_SIZE, (1400.,-1800.,900.), [(8,8,8),(4,4,32)], 640)
run("stress", 10, (160,160,40), syn.SPACE_SIZE, syn.SPACE_CENTER, [(8,8,8),(4,4,40),(8,8,4)], 640)
run("coarse", 5, (80,80,20), syn.SPACE_SIZE, syn.SPACE_CENTER, [(8,8,4),(4,4,20)], 640)
