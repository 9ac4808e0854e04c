"""Independent NumPy/SciPy restatement of SE(3) used only to cross-check the oracle.

Matrix-exponential based (scipy.linalg.expm/logm on 4x4 homogeneous matrices) so that
it shares no closed-form code with oracle/ilqr_oracle.c.  Poses are [t ; q(w,x,y,z)].
"""
import numpy as np
from scipy.linalg import expm, logm
from scipy.spatial.transform import Rotation


def hat3(a):
    return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0.0]])


def hat6(tau):
    M = np.zeros((4, 4))
    M[:3, :3] = hat3(tau[3:])
    M[:3, 3] = tau[:3]
    return M


def vee6(M):
    return np.array([M[0, 3], M[1, 3], M[2, 3], M[2, 1], M[0, 2], M[1, 0]])


def pose_to_mat(T):
    T = np.asarray(T, dtype=float)
    M = np.eye(4)
    M[:3, :3] = Rotation.from_quat([T[4], T[5], T[6], T[3]]).as_matrix()
    M[:3, 3] = T[:3]
    return M


def mat_to_pose(M):
    q = Rotation.from_matrix(M[:3, :3]).as_quat()  # x y z w
    return np.concatenate([M[:3, 3], [q[3], q[0], q[1], q[2]]])


def exp_mat(tau):
    return expm(hat6(np.asarray(tau, dtype=float)))


def log_mat(M):
    return vee6(np.real(logm(M)))


def rminus_mat(Y, X):
    """Y (-) X = Log(X^-1 Y) on 4x4 matrices"""
    return log_mat(np.linalg.inv(X) @ Y)


def same_rotation(q1, q2, tol):
    """quaternions equal up to sign"""
    q1, q2 = np.asarray(q1), np.asarray(q2)
    return min(np.linalg.norm(q1 - q2), np.linalg.norm(q1 + q2)) < tol
