def params_html_table(params):
    return "<pre>{0}</pre>".format(repr(params))
