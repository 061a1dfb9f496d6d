"""Cost helpers.  quaternion_to_yaw mirrors reference mppiisaac/utils/conversions.py:4-11
(xyzw quaternions, batch [K,4] -> yaw [K]); pinned by tests/golden/quaternion_to_yaw.json."""
import torch


def quaternion_to_yaw(quat: torch.Tensor) -> torch.Tensor:
    if getattr(quat, "_mppi_sym", False):   # (an Objective being traced into a cost program: mppiisaac/trace.py)
        from mppiisaac import trace
        return trace.quaternion_to_yaw(quat)
    x, y, z, w = quat[:, 0], quat[:, 1], quat[:, 2], quat[:, 3]
    siny_cosp = 2.0 * (w * z + x * y)
    cosy_cosp = w * w + x * x - y * y - z * z
    return torch.atan2(siny_cosp, cosy_cosp)


def quaternion_to_matrix(quaternions: torch.Tensor) -> torch.Tensor:
    """pytorch3d.transforms.quaternion_to_matrix (0.3.0; real-first (r,i,j,k) convention).  The
    reference examples call it on xyzw rows (examples/panda/planner.py:30-32); shipped here because
    pytorch3d is not a dependency of this backend."""
    if getattr(quaternions, "_mppi_sym", False):
        from mppiisaac import trace
        return trace.quaternion_to_matrix(quaternions)
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    o = torch.stack(
        (1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
         two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
         two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(quaternions.shape[:-1] + (3, 3))


def _angle_from_tan(axis: str, other_axis: str, data, horizontal: bool, tait_bryan: bool):
    i1, i2 = {"X": (2, 1), "Y": (0, 2), "Z": (1, 0)}[axis]
    if horizontal:
        i2, i1 = i1, i2
    even = (axis + other_axis) in ["XY", "YZ", "ZX"]
    if horizontal == even:
        return torch.atan2(data[..., i1], data[..., i2])
    if tait_bryan:
        return torch.atan2(-data[..., i2], data[..., i1])
    return torch.atan2(data[..., i2], -data[..., i1])


def matrix_to_euler_angles(matrix: torch.Tensor, convention: str) -> torch.Tensor:
    """pytorch3d.transforms.matrix_to_euler_angles (0.3.0), e.g. convention "ZYX"."""
    if getattr(matrix, "_mppi_sym", False):
        from mppiisaac import trace
        return trace.matrix_to_euler_angles(matrix, convention)
    idx = {"X": 0, "Y": 1, "Z": 2}
    i0, i2 = idx[convention[0]], idx[convention[2]]
    tait_bryan = i0 != i2
    if tait_bryan:
        central = torch.asin(matrix[..., i0, i2] * (-1.0 if i0 - i2 in [-1, 2] else 1.0))
    else:
        central = torch.acos(matrix[..., i0, i0])
    o = (_angle_from_tan(convention[0], convention[1], matrix[..., i2], False, tait_bryan),
         central,
         _angle_from_tan(convention[2], convention[1], matrix[..., i0, :], True, tait_bryan))
    return torch.stack(o, -1)
