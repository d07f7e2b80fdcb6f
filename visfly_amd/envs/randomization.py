"""State randomizers of the env layer (reference: utils/randomization.py, envs/base/droneEnv.py:145-251).

Two consumers:
  * the fused HIP env kernel spawns on the device from ``vf_spawn_box`` records
    (``spawn_boxes`` turns the reference's ``random_kwargs["state_generator"]`` tree into them);
  * parity mode replays the reference's host RNG stream: ``ReplaySpawner`` draws from one
    ``torch.Generator`` in exactly the order the reference consumes its global generator
    (SURVEY App. B.3), so reset states come out bit-identical.
"""
from typing import Dict, List, Optional

import torch as th

_ZERO = {"mean": [0., 0., 0.], "half": [0., 0., 0.]}
_FIELDS = ("position", "orientation", "velocity", "angular_velocity")


def _uniform_box(kwargs: Dict) -> Dict[str, Dict[str, List[float]]]:
    for k in kwargs:
        if k not in _FIELDS:
            raise NotImplementedError(
                f"state generator option '{k}' (heading/test grids need the scene collision query) is not "
                "part of the MI355X hot path")
    return {f: {"mean": [float(x) for x in kwargs.get(f, _ZERO)["mean"]],
                "half": [float(x) for x in kwargs.get(f, _ZERO)["half"]]} for f in _FIELDS}


def spawn_boxes(random_kwargs: Optional[Dict]) -> List[Dict]:
    """random_kwargs["state_generator"] -> list of uniform boxes (1 = Uniform, >1 = Union)
    (droneEnv.py:145-150,221-230; randomization.py:108-170,250-296)"""
    sg = (random_kwargs or {}).get("state_generator", {})
    cls = sg.get("class", "Uniform")
    kwargs_list = sg.get("kwargs", [{}])
    if cls == "Uniform":
        return [_uniform_box(kwargs_list[0])]
    if cls == "Union":
        boxes = []
        for r in kwargs_list[0]["randomizers_kwargs"]:
            if r["class"] != "Uniform":
                raise NotImplementedError("Union members other than Uniform are not supported")
            boxes.append(_uniform_box(r["kwargs"]))
        return boxes
    raise NotImplementedError(f"state generator class '{cls}' is not part of the MI355X hot path "
                              "(Normal / TargetUniform need features outside SURVEY.md 8)")


def _from_euler(roll, pitch, yaw, cr_trig=False):
    """Quaternion.from_euler, zyx (utils/maths.py:256-269); same torch CPU ops -> same bits as the reference run by this torch
    build.  cr_trig: sin / cos evaluated in fp64 and rounded once instead, as the golden generator's CR-trig patch evaluates them
    (oracle/gen_golden.py::use_cr_trig) -- only for reproducing the CR-patched fixtures; differs from torch's own sin / cos by one
    ulp on a few per cent of the arguments"""
    cos = (lambda x: th.cos(x.double()).float()) if cr_trig else th.cos
    sin = (lambda x: th.sin(x.double()).float()) if cr_trig else th.sin
    cy, sy = cos(yaw * 0.5), sin(yaw * 0.5)
    cp, sp = cos(pitch * 0.5), sin(pitch * 0.5)
    cr, sr = cos(roll * 0.5), sin(roll * 0.5)
    return th.stack([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy,
                     cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy], dim=1)


class ReplaySpawner:
    """Host-side replay of the reference's spawn draws on a shared torch CPU generator."""

    def __init__(self, boxes: List[Dict], rng: th.Generator, cr_trig: bool = False):
        self.rng, self.cr_trig = rng, cr_trig
        self.boxes = [{f: (th.tensor(b[f]["mean"]), th.tensor(b[f]["half"])) for f in _FIELDS} for b in boxes]

    def _uniform(self, box, num):
        """UniformStateRandomizer._generate (randomization.py:153-170): 4 draws of (num,3)"""
        out = []
        for f in _FIELDS:
            mean, half = box[f]
            u = 2 * th.rand(num, 3, generator=self.rng) - 1
            if f == "position":
                out.append(mean.unsqueeze(0) + u * half.unsqueeze(0))
            else:
                out.append(u * half.unsqueeze(0) + mean.unsqueeze(0))
        return out

    def generate(self, num: int):
        """-> pos (num,3), quat (num,4), vel (num,3), ang_vel (num,3); the reference loops over agents
        calling safe_generate(num=1) (droneEnv.py:243-249), which consumes the stream agent by agent."""
        if len(self.boxes) == 1:
            # bulk draw == the per-agent th.rand(1,3) sequence (verified: torch CPU MT19937 stream is
            # consumed row-major), so one (num,4,3) draw replaces the reference's O(N) Python loop
            u = 2 * th.rand(num, 4, 3, generator=self.rng) - 1
            box = self.boxes[0]
            (pm, ph), (om, oh), (vm, vh), (wm, wh) = (box[f] for f in _FIELDS)
            p = pm.unsqueeze(0) + u[:, 0] * ph.unsqueeze(0)
            o = u[:, 1] * oh.unsqueeze(0) + om.unsqueeze(0)
            v = u[:, 2] * vh.unsqueeze(0) + vm.unsqueeze(0)
            w = u[:, 3] * wh.unsqueeze(0) + wm.unsqueeze(0)
            return p, _from_euler(o[:, 0], o[:, 1], o[:, 2], self.cr_trig), v, w
        rows = []
        for _ in range(num):  # UnionRandomizer._generate (:284-296): every member draws, then randint picks
            members = [self._uniform(b, 1) for b in self.boxes]
            sel = int(th.randint(0, len(self.boxes), (1,), generator=self.rng))
            p, o, v, w = members[sel]
            rows.append((p, _from_euler(o[:, 0], o[:, 1], o[:, 2], self.cr_trig), v, w))
        cat = lambda k: th.cat([r[k] for r in rows]) if rows else th.zeros((0, (3, 4, 3, 3)[k]))
        return cat(0), cat(1), cat(2), cat(3)
