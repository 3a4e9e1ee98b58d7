"""Drop-in for the reference's `models.poses` (script/models/poses.py): LearnPose without lietorch."""
from nefes_amd.pose import LearnPose, make_c2w, se3_exp, so3_exp  # noqa: F401
