"""advmix_amd.dataset - host mirror of the reference's lib/dataset entry points on the MI355X path."""
