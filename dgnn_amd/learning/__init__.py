"""Drop-in mirrors of the reference's ``learning/`` model modules (same file names, class names,
constructor arguments, method names and state_dict keys), executing on the HIP kernels."""
