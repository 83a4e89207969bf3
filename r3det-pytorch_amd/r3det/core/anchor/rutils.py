"""ranchor_inside_flags (core/anchor/rutils.py:1-29): which rotated anchors may be assigned."""


def ranchor_inside_flags(flat_ranchors, valid_flags, img_shape, allowed_border=0):
    """With ``allowed_border >= 0`` an anchor counts when its CENTRE lies within the image grown by
    the border (:19-26); the shipped configs pass -1, which keeps ``valid_flags`` as they are."""
    if allowed_border < 0:
        return valid_flags
    img_h, img_w = img_shape[:2]
    cx, cy = flat_ranchors[:, 0], flat_ranchors[:, 1]
    return (valid_flags & (cx >= -allowed_border) & (cy >= -allowed_border)
            & (cx < img_w + allowed_border) & (cy < img_h + allowed_border))
