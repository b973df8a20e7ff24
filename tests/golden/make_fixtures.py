"""Regenerates the committed fixtures under tests/golden/ (run in the build container only: it reads
/root/reference, which does not exist on the GPU box).

  test_yahr_scene.yahr   the exact scene text of the reference's end-to-end test (compat/test_yahr.py:11-52),
                         produced by IMPORTING the reference's own serialiser compat/yahr.py -- this pins what
                         text the loader must accept (bare negative literals, `(BVH 16 Midpoint)`, %d focal).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "compat"))
import yahr as ref  # the reference's scene writer (pure Python)

scene = ref.Scene(
    ref.WhittedIntegrator(3), ref.BVH(16),
    ref.Camera(imW=10, imH=10, focalLength=5, lookDir=ref.Vec3(0, -0.1, -1), upDir=ref.Vec3(0, 1, 0),
               position=ref.Vec3(0, 0.2, 1)),
    [ref.BlinnPhongMaterial(id="testmat", ambient=ref.Vec3(0, 0, 0), diffuse=ref.Vec3(0.8, 0.8, 0.8),
                            specular=ref.Vec3(0, 0, 0), shininess=1)],
    [ref.PointLight(ref.Vec3(4, 10, 10), ref.Vec3(1, 1, 1))],
    [ref.TriangleMesh(
        [ref.Vec3(-1000, -2.2, -1000), ref.Vec3(1000, -2.2, -1000), ref.Vec3(1000, -2.2, 1000), ref.Vec3(-1000, -2.2, 1000)],
        [ref.Vec3(0, 1, 0)] * 4, [(0, 1, 2), (0, 2, 3)], [True, False], "testmat")])
with open(os.path.join(HERE, "test_yahr_scene.yahr"), "w") as f:
    f.write(scene.repr())
print("wrote fixtures")
