"""Grasp-and-lift experiment on the CPU oracle (fp64): the 40-mm, 1-gram block of panda_pick between the open fingers,
fingers closed by their velocity drives, then the arm lifts.  python tools/exp/grasp_kat.py [explicit]"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mppi-isaac_amd'); sys.path.insert(0, '/root/repo/tests')
from scenes import panda_pick
from oracle.oracle import Oracle
from mppiisaac.backend import capi

o = Oracle("f64")
scene, m, cfg, cost, dof, root = panda_pick(K=32, H=12)
if "explicit" in sys.argv:
    m.contact_flags |= 2
q, qd, ro = dof[0::2].astype(float).copy(), dof[1::2].astype(float).copy(), root.astype(float).copy()
blk = scene.actor_index("panda_pick_block")
q[7] = q[8] = float(next((a.split('=')[1] for a in sys.argv if a.startswith('open=')), 0.0205))   # fingers open
rb, _ = o.rigid_body_state(m, ro, q, qd)
lf, rf = scene.rigid_body_index("panda", "panda_leftfinger"), scene.rigid_body_index("panda", "panda_rightfinger")
mid = 0.5 * (rb[lf, :3] + rb[rf, :3])
print("finger frames", rb[lf, :3], rb[rf, :3])
# finger pads: the box centre is 27 mm along the finger's z (pointing down here); block centred between the pads
hand = scene.rigid_body_index("panda", "panda_hand")
from scipy.spatial.transform import Rotation as Rot
Rl = Rot.from_quat(rb[lf, 3:7]).as_matrix()
pad = mid + Rl @ np.array([0, 0, 0.035])
ro[blk, :3] = pad
ro[blk, 3:7] = rb[lf, 3:7]   # block axes along the finger axes
ro[blk, 7:13] = 0
print("block at", ro[blk, :3], "table top 0.14")
u = np.zeros(9)
hist = []
def run(n, u, tag):
    global ro, q, qd
    for i in range(n):
        ro, q, qd, cf = o.scene_step(m, ro, q, qd, u)
        rbs, _ = o.rigid_body_state(m, ro, q, qd)
        mid = 0.5 * (rbs[lf, :3] + rbs[rf, :3]) + Rot.from_quat(rbs[lf, 3:7]).as_matrix() @ np.array([0, 0, 0.035])
        print(f"{tag} {i:3d} fingers {q[7]*1e3:7.3f} {q[8]*1e3:7.3f} mm  qd {qd[7]:7.4f} {qd[8]:7.4f}  block-pad {np.round((ro[blk,:3]-mid)*1e3,3)} mm  vblk {np.round(ro[blk,7:10],4)} w {np.abs(ro[blk,10:13]).max():.3f} cf_blk {np.round(cf[scene.rigid_body_index('panda_pick_block','box')],3)}")
u[7] = u[8] = -0.1
run(8, u, "close")
u[:] = 0; u[7] = u[8] = -0.1
u[1] = -0.3   # shoulder lifts
run(16, u, "lift ")
u[:] = 0; u[7] = u[8] = -0.1
run(20, u, "hold ")
