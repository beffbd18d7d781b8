class RRTMGLongwave:  # placeholder, replaced below in this round
    pass
