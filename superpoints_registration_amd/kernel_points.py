"""Kernel-point dispositions for KPConv.

Mirrors ``load_kernels`` of the reference
(src/models/backbone_kpconv/kernels/kernel_points.py:387-469).  The reference
reads the unit-radius 15-point 'center' disposition from the binary PLY file
``src/kernels/dispositions/k_015_center_3D.ply`` (cwd-relative lookup, :390),
applies a random rotation about z, adds N(0, 0.01) noise, scales by the
convolution radius and rotates.  The 15 x 3 float64 table below is the decoded
payload of that file (data, not code); other kernel sizes need the offline
optimiser of the reference and are not supported here.

Kernel points are a frozen Parameter in the state dict
(kpconv_blocks.py:266), so a checkpoint always overrides this initialiser.
"""
import numpy as np

K015_CENTER_3D = np.array([
    [0.0, 0.0, 0.0],
    [0.36145941026597067, 0.48239212397030184, -0.27125260188676675],
    [-0.48163767397727025, -0.22902570148057874, 0.3905206150186844],
    [0.4372927010555656, -0.49427528002130244, 0.03741766626651985],
    [-0.5198507833083387, -0.2979149446938551, -0.279168209843934],
    [-0.12344802259344391, -0.6311186121210636, 0.15291476004445942],
    [-0.6096160563430828, 0.24541086682383462, -0.07123770770511689],
    [-0.1689108743398532, 0.635534777653165, -0.06707186148862122],
    [0.6399171252503684, 0.04467887639897198, 0.1595112613255694],
    [-0.17213346075329766, 0.20048248013725176, -0.5981755025792743],
    [0.2818393803824886, 0.44664145088924706, 0.39750599834002404],
    [0.010631422808218766, -0.45118795265919565, -0.48296001481067796],
    [0.17213346082322145, -0.20048248011358744, 0.598175502567091],
    [0.4512246987754341, -0.06093370553907477, -0.47918305004457534],
    [-0.278901328009585, 0.3097981007681546, 0.5130031447903965],
], dtype=np.float64)


def load_kernels(radius, num_kpoints, dimension=3, fixed="center", lloyd=False):
    """kernel_points.py:387-469.  Uses the global numpy RNG exactly like the
    reference (one rand() for theta, then normal(size=(K,3)))."""
    if num_kpoints != 15 or dimension != 3 or fixed != "center":
        raise NotImplementedError(
            "only the shipped 15-point 3-D 'center' disposition is available "
            f"(asked for K={num_kpoints}, dim={dimension}, fixed={fixed})")
    kernel_points = K015_CENTER_3D.copy()
    theta = np.random.rand() * 2 * np.pi
    c, s = np.cos(theta), np.sin(theta)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
    kernel_points = kernel_points + np.random.normal(scale=0.01, size=kernel_points.shape)
    kernel_points = radius * kernel_points
    kernel_points = np.matmul(kernel_points, R)
    return kernel_points.astype(np.float32)
