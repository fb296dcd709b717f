"""advmix_amd.nms - host mirror of the reference's lib/nms entry points on the MI355X path."""
