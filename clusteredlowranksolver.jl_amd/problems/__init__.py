"""Problem generators for the BASELINE configs, written from the mathematics
of the reference's examples/ directory.  Each returns a `ClusteredLowRankSDP`."""
from .polyopt import polyopt, polyopt_random, polyopt_scaled, min_f
from .delsarte import delsarte
from .spherepacking import cohnelkies, nsphere_packing, cohnelkies_multi
from .sdpa import read_sdpa, sdpa_to_sdp, sdpa_scaled, write_sdpa
from .threepoint import three_point_spherical_codes
from .toy import dense_sdp, theta_c5, povm_2x2, toy_z, toy_z_as_free
