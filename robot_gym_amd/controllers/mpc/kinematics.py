"""Host-side leg kinematics behind the controller's `kinematics_model` property.

The reference exposes `kinematics_model.MapContactForceToJointTorques(leg_id, contact_force)` and
`ComputeMotorAnglesFromFootLocalPosition(leg_id, foot_local_position)` (called back through
model/robots/robot.py:94-102) and implements them with pybullet numerics
(controllers/mpc/kinematics.py:13-53,98-133).  Two implementations with that surface:

  ChainKinematics     closed-form 3-revolute chain from the URDF data (chain.json); float64
                      numpy twin of the device code in csrc/rg_mpc_dev.h (leg_fk / leg_ik).
  PybulletKinematics  asks the live pybullet client for the Jacobian like the reference does;
                      used when the controller is dropped into a PyBullet simulation.
"""
import numpy as np


def _rot_rpy(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _rot_axis(axis, ang):
    x, y, z = axis / np.linalg.norm(axis)
    c, s = np.cos(ang), np.sin(ang)
    C = 1 - c
    return np.array([[c + x * x * C, x * y * C - z * s, x * z * C + y * s],
                     [y * x * C + z * s, c + y * y * C, y * z * C - x * s],
                     [z * x * C - y * s, z * y * C + x * s, c + z * z * C]])


class ChainKinematics:
    def __init__(self, cfg):
        self.cfg = cfg
        self.jxyz = np.asarray(cfg.jxyz, dtype=np.float64).reshape(4, 3, 3)
        self.jrpy = np.asarray(cfg.jrpy, dtype=np.float64).reshape(4, 3, 3)
        self.jaxis = np.asarray(cfg.jaxis, dtype=np.float64).reshape(4, 3, 3)
        self.tip = (np.asarray(cfg.toe_xyz, dtype=np.float64) + np.asarray(cfg.toe_com, dtype=np.float64)).reshape(4, 3)
        self.base_com = np.asarray(cfg.base_com, dtype=np.float64)
        self.mdir = np.asarray(cfg.motor_dir, dtype=np.float64)
        self.moff = np.asarray(cfg.motor_off, dtype=np.float64)

    def foot_position_and_jacobian(self, leg_id, motor_angles3):
        """Toe COM in the base frame and the 3x3 joint-space Jacobian d foot / d joint."""
        R, o = np.eye(3), np.zeros(3)
        axes, orgs = [], []
        for j in range(3):
            o = o + R @ self.jxyz[leg_id, j]
            R = R @ _rot_rpy(self.jrpy[leg_id, j])
            ax = self.jaxis[leg_id, j] / np.linalg.norm(self.jaxis[leg_id, j])
            axes.append(R @ ax)
            orgs.append(o.copy())
            m = 3 * leg_id + j
            R = R @ _rot_axis(ax, motor_angles3[j] * self.mdir[m] + self.moff[m])
        pf = o + R @ self.tip[leg_id]
        J = np.stack([np.cross(axes[j], pf - orgs[j]) for j in range(3)], axis=1)
        return pf - self.base_com, J

    def ComputeJacobian(self, leg_id, motor_angles):
        return self.foot_position_and_jacobian(leg_id, np.asarray(motor_angles)[3 * leg_id:3 * leg_id + 3])[1]

    def MapContactForceToJointTorques(self, leg_id, contact_force, motor_angles):
        J = self.ComputeJacobian(leg_id, motor_angles)
        tau = (np.asarray(contact_force, dtype=np.float64) @ J) * self.mdir[3 * leg_id:3 * leg_id + 3]
        return {3 * leg_id + j: tau[j] for j in range(3)}

    def ComputeMotorAnglesFromFootLocalPosition(self, leg_id, foot_local_position, motor_angles):
        cfg = self.cfg
        q = np.asarray(motor_angles, dtype=np.float64)[3 * leg_id:3 * leg_id + 3].copy()
        target = np.asarray(foot_local_position, dtype=np.float64)
        d = self.mdir[3 * leg_id:3 * leg_id + 3]
        for _ in range(cfg.ik_iters):
            p, J = self.foot_position_and_jacobian(leg_id, q)
            Jm = J * d[None, :]
            A = Jm @ Jm.T + cfg.ik_damping * np.eye(3)
            if np.linalg.det(A) == 0.0:
                break
            q += np.clip(Jm.T @ np.linalg.solve(A, target - p), -cfg.ik_max_step, cfg.ik_max_step)
        return list(range(3 * leg_id, 3 * leg_id + 3)), q.tolist()


class PybulletKinematics:
    """Jacobian from the simulator the controller lives in (same calls as the reference adapter)."""

    def __init__(self, robot, chain: ChainKinematics):
        self._robot = robot
        self._chain = chain

    def leg_jacobian(self, leg_id):
        rb = self._robot
        angles = [s[0] for s in rb.GetJointStates]
        zeros = [0] * len(angles)
        jv, _ = rb.pybullet_client.calculateJacobian(rb.GetRobotId, rb.GetFootLinkIds[leg_id], (0, 0, 0), angles, zeros, zeros)
        jv = np.asarray(jv)
        if jv.shape[0] != 3:
            raise ValueError("translational Jacobian must have 3 rows")
        return jv[:, 6 + 3 * leg_id: 9 + 3 * leg_id]

    def all_leg_jacobians(self, out):
        """The four legs' 3 x 3 blocks into out[4, 3, 3] with ONE read of the joint states (the per-tick state gather of the
        drop-in controller: the reference asks per leg and per call, controllers/mpc/kinematics.py:13-30)."""
        rb = self._robot
        angles = [s[0] for s in rb.GetJointStates]
        zeros = [0] * len(angles)
        bullet, rid, links = rb.pybullet_client, rb.GetRobotId, rb.GetFootLinkIds
        for leg_id in range(4):
            jv, _ = bullet.calculateJacobian(rid, links[leg_id], (0, 0, 0), angles, zeros, zeros)
            if len(jv) != 3:
                raise ValueError("translational Jacobian must have 3 rows")
            for i in range(3):
                out[leg_id, i] = jv[i][6 + 3 * leg_id: 9 + 3 * leg_id]

    def MapContactForceToJointTorques(self, leg_id, contact_force):
        J = self.leg_jacobian(leg_id)
        direction = np.asarray(self._robot.GetMotorConstants().MOTOR_DIRECTION, dtype=np.float64)
        tau = (np.asarray(contact_force, dtype=np.float64) @ J) * direction[3 * leg_id:3 * leg_id + 3]
        return {3 * leg_id + j: tau[j] for j in range(3)}

    def ComputeMotorAnglesFromFootLocalPosition(self, leg_id, foot_local_position):
        return self._chain.ComputeMotorAnglesFromFootLocalPosition(leg_id, foot_local_position, self._robot.GetMotorAngles())
