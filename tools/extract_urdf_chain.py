#!/usr/bin/env python3
"""Extract the 3-revolute leg chains from the reference URDFs into small JSON data files.

Run in the authoring container only (needs /root/reference); the JSON it writes is data
(joint origins/axes, toe offsets, link COMs) and is what travels to the GPU box.
Sources: robot_gym/util/pybullet_data/robots/ghost.urdf, k3lso.urdf; motor order from
robot_gym/model/robots/<robot>/marks.py 'motor_names' (FR, FL, RR, RL).
"""
import json
import os
import sys
import xml.etree.ElementTree as ET

REF = os.environ.get("RG_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robot_gym_amd", "model", "robots")


def vec(s, default=(0.0, 0.0, 0.0)):
    return [float(x) for x in s.split()] if s else list(default)


def extract(urdf, motor_names):
    root = ET.parse(urdf).getroot()
    joints = {j.get("name"): j for j in root.findall("joint")}
    links = {l.get("name"): l for l in root.findall("link")}
    child_of = {}
    for j in root.findall("joint"):
        child_of.setdefault(j.find("parent").get("link"), []).append(j)

    def com(link):
        i = links[link].find("inertial")
        o = i.find("origin") if i is not None else None
        return vec(o.get("xyz")) if o is not None else [0.0, 0.0, 0.0]

    base = [l for l in links if all(j.find("child").get("link") != l for j in root.findall("joint"))][0]
    chain = {"base_link": base, "base_com": com(base), "legs": []}
    for leg in range(4):
        names = motor_names[3 * leg:3 * leg + 3]
        jx, jr, ja = [], [], []
        for n in names:
            j = joints[n]
            o = j.find("origin")
            jx.append(vec(o.get("xyz")))
            jr.append(vec(o.get("rpy")))
            ja.append(vec(j.find("axis").get("xyz")))
        lower = joints[names[2]].find("child").get("link")
        toe_joint = [j for j in child_of[lower] if j.get("type") == "fixed"][0]
        to = toe_joint.find("origin")
        assert vec(to.get("rpy")) == [0.0, 0.0, 0.0], "toe joint rotation not supported"
        toe_link = toe_joint.find("child").get("link")
        chain["legs"].append({"joints": names, "xyz": jx, "rpy": jr, "axis": ja,
                              "toe_xyz": vec(to.get("xyz")), "toe_com": com(toe_link), "toe_link": toe_link})
    return chain


def main():
    sys.path.insert(0, REF)
    for robot in ("ghost", "k3lso"):
        marks = {}
        exec(open(os.path.join(REF, "robot_gym/model/robots", robot, "marks.py")).read(), marks)
        mp = marks["MARK_PARAMS"][marks["MARK_LIST"][0]]
        urdf = os.path.join(REF, "robot_gym/util/pybullet_data", mp["urdf_name"])
        chain = extract(urdf, mp["motor_names"])
        chain["source"] = "robot_gym/util/pybullet_data/" + mp["urdf_name"]
        path = os.path.join(OUT, robot, "chain.json")
        with open(path, "w") as f:
            json.dump(chain, f, indent=1)
        print("wrote", path)


if __name__ == "__main__":
    main()
