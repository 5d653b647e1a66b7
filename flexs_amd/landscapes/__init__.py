"""Ground-truth landscapes that are pure table look-ups, moved onto the GPU
(SURVEY.md section 8f-4).  The simulator-backed landscapes of the reference
(ViennaRNA, PyRosetta, TAPE) stay the reference's own."""
from flexs_amd.landscapes.tf_binding import TFBinding  # noqa: F401
from flexs_amd.landscapes.additive_aav_packaging import AdditiveAAVPackaging  # noqa: F401
